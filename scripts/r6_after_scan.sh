#!/bin/bash
# round 6: after the one-pass scan -- the c3 step and the command line on the 100 M-record files
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-coverage --no-dist-leg 2>/dev/null | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac']); pk=d['roofline']['per_kernel']; print('scan', pk.get('scan'))"
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
    for flag in -b -bu; do run now /tmp/$f.bam $flag; done
  done
done
$D digest /tmp/f.bam
