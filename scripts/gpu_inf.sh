#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_cli_scale.py -x -q -m gpu -k "inflated" 2>&1 | tail -4
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
for cfg in MSX_INFLATE_REFUSE=1 MSX_HOST_INFLATE=1; do for i in 1 2; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 $cfg $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "batches:\|on the host" /tmp/err.txt | cut -c1-80 | tr '\n' ' '; cat /tmp/time.txt; done; done
$B digest /tmp/f.bam | tail -1
