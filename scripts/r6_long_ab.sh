#!/bin/bash
# round 6: filter's batches of 4096 blocks (after the first eight) against 2048 throughout, alternating: the 400 M-record file,
# the 100 M-record lean and SEQ/QUAL files
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
run() { # name file flag env...
  rm -f /tmp/f.bam; sleep 0.7
  local a=$EPOCHREALTIME
  env MSX_TIMING=1 "${@:4}" $B filter -l 80 -p 95 -z 80 --besthit $3 --profile-out /tmp/p.gz --label S $2 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1 $3 $(basename $2)] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-200)"
}
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
$D synth --groups 20000000 --refs 1000000 --seq -b > /tmp/seq100.bam
for rep in 1 2 3; do
  for f in lean100 seq100; do
    for flag in -b; do
      run ramp4096 /tmp/$f.bam $flag
      run all2048 /tmp/$f.bam $flag MSX_COMP_BLOCKS=2048
    done
  done
done
rm -f /tmp/seq100.bam /tmp/lean100.bam
$D synth --groups 80000000 --refs 1000000 -b > /tmp/big.bam
for rep in 1 2 3 4; do
  run ramp4096 /tmp/big.bam -b
  run all2048 /tmp/big.bam -b MSX_COMP_BLOCKS=2048
done
$D digest /tmp/f.bam
