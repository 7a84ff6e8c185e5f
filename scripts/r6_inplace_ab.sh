#!/bin/bash
# round 6: a batch sent ahead walked where it was inflated (default) against copied behind the carry (debug library, MSX_UP_HEAD=256)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
run() { # name file flag env...
  rm -f /tmp/f.bam; sleep 0.7
  local a=$EPOCHREALTIME
  env MSX_TIMING=1 "${@:4}" $B filter -l 80 -p 95 -z 80 --besthit $3 --profile-out /tmp/p.gz --label S $2 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1 $3 $(basename $2)] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-200)"
}
DBG=$GRAFT_REPO_ROOT/msamtools_amd/dbg
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
$D synth --groups 20000000 --refs 1000000 --seq -b > /tmp/seq100.bam
for rep in 1 2 3 4; do
  for f in lean100 seq100; do
    for flag in -b -bu; do
      run inplace /tmp/$f.bam $flag LD_LIBRARY_PATH=$DBG
      run copy /tmp/$f.bam $flag LD_LIBRARY_PATH=$DBG MSX_UP_HEAD=256
    done
  done
done
$D digest /tmp/f.bam
