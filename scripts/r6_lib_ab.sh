#!/bin/bash
# round 6: two builds of the library side by side under the command line (msamtools_amd/alt/ first on LD_LIBRARY_PATH)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
run() { # name file flag env...
  rm -f /tmp/f.bam; sleep 0.7
  local a=$EPOCHREALTIME
  env MSX_TIMING=1 "${@:4}" $B filter -l 80 -p 95 -z 80 --besthit $3 --profile-out /tmp/p.gz --label S $2 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1 $3 $(basename $2)] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-200) $(ls -l /tmp/f.bam | awk '{print $5}')"
}
ALT=$GRAFT_REPO_ROOT/msamtools_amd/alt
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
$D synth --groups 20000000 --refs 1000000 --seq -b > /tmp/seq100.bam
for rep in 1 2 3; do
  for f in lean100 seq100; do
    run base /tmp/$f.bam -b
    run alt /tmp/$f.bam -b LD_LIBRARY_PATH=$ALT
  done
done
rm -f /tmp/lean100.bam /tmp/seq100.bam
$D synth --groups 80000000 --refs 1000000 -b > /tmp/big.bam
for rep in 1 2 3; do
  run base /tmp/big.bam -b
  run alt /tmp/big.bam -b LD_LIBRARY_PATH=$ALT
done
$D digest /tmp/f.bam
