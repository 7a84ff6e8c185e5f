#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R"
for cfg in X=1 MSX_NO_PIN=1 MSX_NO_MMAP=1 "MSX_SLOTS=3" X=2; do echo "== $cfg"; for i in 1 2 3; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 $cfg $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; pr=$(grep "process:" /tmp/err.txt | sed 's/# process: \([0-9.]*\) s.*/\1/'); re=$(cat /tmp/time.txt); echo "process $pr real $re outside $(python3 -c "print(round($re-$pr,3))")"; done; done
