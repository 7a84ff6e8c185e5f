#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
timeout 900 python -m pytest tests/test_cli_scale.py tests/test_host_cli.py -x -q -m gpu -k "coverage" 2>&1 | tail -2
$B synth --groups 10000000 --refs 50000 -b > /tmp/in50k.bam
TIMEFORMAT="%R s real"
echo "== per-position text, 50k refs"; for i in 1 2; do { time env MSX_TIMING=1 $B coverage -o /tmp/t.gz /tmp/in50k.bam 2> /tmp/err.txt; } 2>&1 | tail -1; grep "# coverage" /tmp/err.txt; done; ls -l /tmp/t.gz; zcat /tmp/t.gz | md5sum
{ time $B coverage -x -w 40 -o - /tmp/in50k.bam 2>/dev/null | zcat | md5sum; } 2>&1 | tail -2
{ time env MSX_SERIAL_IO=1 MSX_THREADS=1 $B coverage -x -w 40 -o - /tmp/in50k.bam 2>/dev/null | zcat | md5sum; } 2>&1 | tail -2
