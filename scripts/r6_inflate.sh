#!/bin/bash
# round 6: the lane-parallel inflater -- its tests (both kernels), then rate, clocks per phase and every block against zlib on a
# lean and a SEQ/QUAL file and on a header's blocks: gpurun -- 'bash scripts/r6_inflate.sh NAME [blocks]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=${1:-r6_inflate}; NB=${2:-8192}
OUT=gpurun_out/$NAME; rm -rf $OUT; mkdir -p $OUT
DEV=msamtools_amd/bin/msamtools-dev
timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu --timeout=240 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
$DEV synth --groups 1800000 --refs 100000 -b > /tmp/lean.bam
$DEV synth --groups 500000 --refs 100000 --seq -b > /tmp/seq.bam
for f in lean seq; do
  MSX_INFLATE_STATS=3 timeout 600 python scripts/bench_inflate.py /tmp/$f.bam $NB --skip-bytes 3000000 2>&1 | grep phases | head -1 | tee $OUT/$f.phases.log
  MSX_INFLATE_STATS=2 timeout 600 python scripts/bench_inflate.py /tmp/$f.bam $NB --skip-bytes 3000000 --json $OUT/$f.json 2>&1 | tee $OUT/$f.log | grep -v "^#\|handed back to the serial kernel, 0 refused" | tail -3
  grep "handed back" $OUT/$f.log | sort | uniq -c | tail -2
done
MSX_INFLATE_STATS=3 timeout 600 python scripts/bench_inflate.py /tmp/lean.bam 64 2>&1 | grep phases | head -1 | tee $OUT/header.phases.log
MSX_INFLATE_STATS=2 timeout 600 python scripts/bench_inflate.py /tmp/lean.bam 64 --json $OUT/header.json 2>&1 | tee $OUT/header.log | tail -4
