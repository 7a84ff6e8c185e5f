#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
run() { for i in 1 2 3; do { time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "filter pipeline" /tmp/err.txt | sed 's/; decode/ decode/; s/ device: start-up.*encode/ encode/; s/99992794 records.*//' | tr '\n' ' '; cat /tmp/time.txt; sleep 2; done; }
for t in 16 20 24 28; do echo "== MSX_THREADS=$t"; run MSX_THREADS=$t; done
