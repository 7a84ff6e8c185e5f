#!/bin/bash
# round 6: the lane-parallel inflater's clocks per phase on a lean and a SEQ/QUAL file and on a header's blocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=${1:-r6_phases}; NB=${2:-8192}
OUT=gpurun_out/$NAME; rm -rf $OUT; mkdir -p $OUT
DEV=msamtools_amd/bin/msamtools-dev
$DEV synth --groups 1800000 --refs 100000 -b > /tmp/lean.bam
$DEV synth --groups 500000 --refs 100000 --seq -b > /tmp/seq.bam
for f in lean seq; do
  MSX_INFLATE_STATS=3 timeout 600 python scripts/bench_inflate.py /tmp/$f.bam $NB --skip-bytes 3000000 2>&1 | grep "phases\|lanes" | sed -n '2p;$p' | tee $OUT/$f.phases.log
done
MSX_INFLATE_STATS=3 timeout 600 python scripts/bench_inflate.py /tmp/lean.bam 64 2>&1 | grep "phases\|lanes" | sed -n '2p;$p' | tee $OUT/header.phases.log
