#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
time $B synth --groups 10000000 --refs 1000000 -b --seq > /tmp/seq.bam
ls -l /tmp/seq.bam
TIMEFORMAT="%R s real"
for i in 1 2 3; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/seq.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "filter pipeline\|process:" /tmp/err.txt | sed 's/ [0-9]* records in.*//; s/, [0-9.]* s of CPU.*//' | tr '\n' ' '; cat /tmp/time.txt; done
