#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2
B=msamtools_amd/bin/msamtools
$B synth --groups 2000000 --refs 1000000 -b > /tmp/in.bam
mkdir -p gpurun_out/inf2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf2 -o t -- python3 scripts/bench_inflate.py /tmp/in.bam 8192 2>&1 | tail -1
find gpurun_out/inf2 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-60,150-230 | head -4
for n in 2048; do timeout 300 python scripts/bench_inflate.py /tmp/in.bam $n 2>&1 | tail -1; done
