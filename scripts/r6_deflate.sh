#!/bin/bash
# round 6: the encoder -- its tests (blocks equal to the host twin's bit for bit), then rate and clocks per phase on lean and SEQ/QUAL records
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_deflate}; rm -rf $OUT; mkdir -p $OUT
DEV=msamtools_amd/bin/msamtools-dev
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q -m gpu --timeout=300 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
# (inputs of several rounds of waves: 15 000 blocks and more)
$DEV synth --groups 3200000 --refs 100000 -u > /tmp/lean_u.bam
$DEV synth --groups 1200000 --refs 100000 --seq -u > /tmp/seq_u.bam
for f in lean seq; do
  MSX_DEFLATE_STATS=1 timeout 600 python scripts/bench_deflate.py /tmp/${f}_u.bam 2>&1 | tee $OUT/$f.log | grep -v "^# deflate: geometry" | tail -4 | cut -c1-500
done
