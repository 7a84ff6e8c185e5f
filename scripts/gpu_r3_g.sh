#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
run() { echo "== $*"; ( time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep "filter pipeline\|^real\|process:"; sleep 2; }
run X=1
run X=1
run MSX_SLOTS=3
run MSX_SLOTS=6
echo "== profile alone"; ( time MSX_TIMING=1 $B profile --label S -o /tmp/p1.gz /tmp/in.bam ) 2>&1 | grep "profile pipeline\|^real\|process:"; sleep 2
echo "== filter alone"; ( time MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep "filter pipeline\|^real\|process:"; sleep 2
echo "== pipe"; ( time sh -c "MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/in.bam | MSX_TIMING=1 $B profile --label S -o /tmp/p2.gz -" ) 2>&1 | grep "pipeline\|^real\|process:"
$B digest /tmp/f.bam
