#!/bin/bash
# round 6: the command line without the context's two side lanes (its default) against with them (MSX_SERIAL=0), alternating on one box
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
for rep in 1 2 3 4 5; do
  for f in lean100 seq100; do
    run nolanes /tmp/$f.bam -b
    run lanes /tmp/$f.bam -b MSX_SERIAL=0
  done
done
