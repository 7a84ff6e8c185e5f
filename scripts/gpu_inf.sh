#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
timeout 1500 python -m pytest tests/test_cli_scale.py tests/test_host_cli.py tests/test_gpu_unpack.py -x -q -m gpu 2>&1 | tail -8
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
run() { for i in 1 2 3; do { time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "batches:\|filter pipeline\|on the host" /tmp/err.txt | sed 's/; decode/ decode/; s/99992794 records.*//' | tr '\n' ' '; cat /tmp/time.txt; $B digest /tmp/f.bam | tail -1; sleep 1; done; }
for cfg in X=1 MSX_HOST_INFLATE=1 MSX_COMP_BLOCKS=4096; do echo "== $cfg"; run $cfg; done
