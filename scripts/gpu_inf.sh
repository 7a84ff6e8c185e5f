#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
run() { for i in 1 2 3 4; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "batches:" /tmp/err.txt | cut -c1-40 | tr '\n' ' '; grep "filter pipeline\|process:" /tmp/err.txt | sed 's/; decode.*//; s/, [0-9.]* s of CPU.*//' | tr '\n' ' '; cat /tmp/time.txt; done; }
for cfg in X=1 MSX_INFLATE_AHEAD=2 X=2; do echo "== $cfg"; run $cfg; done
