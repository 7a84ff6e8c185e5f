#!/bin/bash
# where the fixed cost of one command line goes: a tiny input with a small and with a 1 M-reference header, timed from
# outside and from inside (MSX_TIMING)
cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
for refs in 1000 1000000; do
  $D synth --groups 2000 --refs $refs -b > /tmp/tiny.bam
  ls -l /tmp/tiny.bam | awk '{print "input bytes", $5}'
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/tiny.bam > /tmp/f.bam 2> /tmp/err.log
    t1=$(date +%s.%N)
    python3 -c "print('refs $refs filter+profile: outside', round($t1-$t0,3), 's')"
    grep -E "filter pipeline|writer done" /tmp/err.log | cut -c1-400
  done
  t0=$(date +%s.%N); $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/tiny.bam > /tmp/f.bam 2>/dev/null; t1=$(date +%s.%N); python3 -c "print('refs $refs filter alone:', round($t1-$t0,3), 's')"
  t0=$(date +%s.%N); $B profile --label S -o /tmp/p.gz /tmp/tiny.bam 2>/dev/null; t1=$(date +%s.%N); python3 -c "print('refs $refs profile alone:', round($t1-$t0,3), 's')"
  t0=$(date +%s.%N); $D digest /tmp/tiny.bam > /dev/null; t1=$(date +%s.%N); python3 -c "print('refs $refs digest (header + records, no GPU):', round($t1-$t0,3), 's')"
done
