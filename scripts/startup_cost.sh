#!/bin/bash
# where the fixed cost of one command line goes: a tiny input, timed from outside and from inside
cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 2000 --refs 1000 -b > /tmp/tiny.bam
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/tiny.bam > /tmp/f.bam 2> /tmp/err.log
  t1=$(date +%s.%N)
  python3 -c "print('tiny filter+profile: outside', round($t1-$t0,3), 's')"
  grep -E "filter pipeline|batch 0" /tmp/err.log | cut -c1-330
done
t0=$(date +%s.%N); $B help > /dev/null 2>&1; t1=$(date +%s.%N); python3 -c "print('msamtools help (load + exit):', round($t1-$t0,3), 's')"
t0=$(date +%s.%N); /bin/true; t1=$(date +%s.%N); python3 -c "print('/bin/true:', round($t1-$t0,3), 's')"
LD_DEBUG=statistics $B help 2>&1 | grep -E "total startup time|relocation|load" | head -5
ldd $B | wc -l
