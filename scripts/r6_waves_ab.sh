#!/bin/bash
# round 6: how many waves per compute unit the inflater and the encoder hold while a batch's small kernels run beside them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
run() { # name file flag env...
  rm -f /tmp/f.bam; sleep 0.7
  local a=$EPOCHREALTIME
  env MSX_TIMING=1 "${@:4}" $B filter -l 80 -p 95 -z 80 --besthit $3 --profile-out /tmp/p.gz --label S $2 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1 $3 $(basename $2)] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-200)"
}
$D synth --groups 20000000 --refs 1000000 --seq -b > /tmp/seq100.bam
$D synth --groups 80000000 --refs 1000000 -b > /tmp/big.bam
for rep in 1 2 3; do
  for f in big seq100; do
    run default /tmp/$f.bam -b
    run inf8 /tmp/$f.bam -b MSX_INFLATE_WAVES=8
    run inf12 /tmp/$f.bam -b MSX_INFLATE_WAVES=12
    run def7 /tmp/$f.bam -b MSX_DEFLATE_WAVES=7
    run def6inf8 /tmp/$f.bam -b MSX_DEFLATE_WAVES=6 MSX_INFLATE_WAVES=8
  done
done
