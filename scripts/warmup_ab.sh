#!/bin/bash
# A/B of the runtime warm-up thread (MSX_NO_WARMUP=1: off) on the 100 M-record command, output discarded, alternating runs
cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
for rep in 1 2 3 4 5 6 7 8; do
  for e in X=1 MSX_NO_WARMUP=1; do
    t0=$(date +%s.%N)
    env MSX_TIMING=1 $e $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /dev/null 2> /tmp/err.log
    t1=$(date +%s.%N)
    echo "$e $(python3 -c "print(round($t1-$t0,3))") $(grep '# process' /tmp/err.log | cut -c12-17) $(grep 'filter pipeline' /tmp/err.log | sed 's/.*start-up \([0-9.]*\),.*/\1/')"
  done
done | python3 -c "
import sys, statistics as st
d={}
for l in sys.stdin:
    e,o,p,s=l.split(); d.setdefault(e,[]).append((float(o),float(p),float(s)))
for e,v in d.items():
    print(e, 'outside median', st.median(x[0] for x in v), 'min', min(x[0] for x in v), '| main..exit median', st.median(x[1] for x in v), '| start-up median', st.median(x[2] for x in v), [x[0] for x in v])
"
