#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
timeout 900 python -m pytest tests/test_cli_scale.py -x -q -m gpu 2>&1 | tail -2
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
for i in 1 2 3 4 5; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "filter pipeline\|process:" /tmp/err.txt | sed 's/; decode.*//; s/, [0-9.]* s of CPU.*//' | tr '\n' ' '; cat /tmp/time.txt; done
$B digest /tmp/f.bam | tail -1
