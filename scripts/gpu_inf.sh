#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -3
B=msamtools_amd/bin/msamtools
$B synth --groups 2000000 --refs 1000000 -b > /tmp/in.bam
for n in 2048 8192; do timeout 300 python scripts/bench_inflate.py /tmp/in.bam $n 2>&1 | tail -1; done
