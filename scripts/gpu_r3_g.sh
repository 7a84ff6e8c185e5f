#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
run() { echo "== $*"; ( time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep "filter pipeline\|^real\|process:"; sleep 1; }
run X=1
run X=1
run MSX_SLOTS=4
run MSX_SLOTS=5
run MSX_SLOTS=5 MSX_BATCH_BYTES=50000000
run MSX_SLOTS=6 MSX_BATCH_BYTES=33554432
echo "== to /dev/null"; ( time MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /dev/null ) 2>&1 | grep "filter pipeline\|^real\|process:"; sleep 1
echo "== pipe"; ( time sh -c "MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/in.bam | MSX_TIMING=1 $B profile --label S -o /tmp/p2.gz -" ) 2>&1 | grep "pipeline\|^real\|process:"
