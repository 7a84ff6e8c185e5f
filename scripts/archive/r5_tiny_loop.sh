#!/bin/bash
# the command line on inputs small enough for the warm-up thread and the device thread to meet inside the runtime's module
# loading (round 5: "Cannot find Symbol" aborts, twice in ~400 command-line test runs): N runs each of four tiny commands,
# counting every exit status that is not the expected one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5tiny; rm -rf $OUT; mkdir -p $OUT
B=msamtools_amd/bin/msamtools; D=msamtools_amd/bin/msamtools-dev
N=${1:-300}
$D recode -b tests/golden/fixtures/besthit.sam > /tmp/t1.bam
$D recode -b tests/golden/fixtures/profile.sam > /tmp/t2.bam
bad=0
for i in $(seq 1 $N); do
  $B filter --besthit -bu /tmp/t1.bam > /dev/null 2> /tmp/e1.txt; r1=$?
  $B profile --label S -o /tmp/p.gz /tmp/t2.bam > /dev/null 2> /tmp/e2.txt; r2=$?
  $B coverage --summary -o /tmp/c.gz /tmp/t1.bam > /dev/null 2> /tmp/e3.txt; r3=$?
  MSX_DEVICES=0,0 $B filter --besthit -bu --profile-out /tmp/p2.gz --label S /tmp/t1.bam > /dev/null 2> /tmp/e4.txt; r4=$?
  if [ "$r1$r2$r3$r4" != "0000" ]; then bad=$((bad+1)); echo "run $i: $r1 $r2 $r3 $r4"; tail -2 /tmp/e1.txt /tmp/e2.txt /tmp/e3.txt /tmp/e4.txt | cut -c1-300; fi
done
echo "runs=$N x 4 commands, bad=$bad" | tee $OUT/summary.txt
