#!/bin/bash
# round 6: every hardware queue the runtime makes costs 173 MB of resident, page-locked host memory (its wave save area): made
# when a stream is first used, taken apart at exit.  GPU_MAX_HW_QUEUES bounds them: fewer queues against less overlap.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
run() { # name file flag env...
  rm -f /tmp/f.bam; sleep 0.7
  local a=$EPOCHREALTIME
  env MSX_TIMING=1 "${@:4}" $B filter -l 80 -p 95 -z 80 --besthit $3 --profile-out /tmp/p.gz --label S $2 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1 $3 $(basename $2)] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-120) $(grep -o 'from main to exit' /tmp/err.log | head -1) $(grep '# process:' /tmp/err.log | cut -c12-19) $(grep -o 'VmRSS: *[0-9]* kB' /tmp/err.log)"
}
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
for rep in 1 2 3; do
  run default /tmp/lean100.bam -b
  for q in 2 3 4 6; do run q$q /tmp/lean100.bam -b GPU_MAX_HW_QUEUES=$q; done
done
$D digest /tmp/f.bam
