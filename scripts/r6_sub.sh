#!/bin/bash
# round 6: bits per lane of the lane-parallel inflater (256 / 512 / 768): builds of the library side by side (MSX_LIB_PATH)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_sub}; rm -rf $OUT; mkdir -p $OUT
DEV=msamtools_amd/bin/msamtools-dev
$DEV synth --groups 1800000 --refs 100000 -b > /tmp/lean.bam
$DEV synth --groups 500000 --refs 100000 --seq -b > /tmp/seq.bam
for lib in _sub320 _sub384 _sub448; do
  for f in lean seq; do
    L=$GRAFT_REPO_ROOT/msamtools_amd/libmsamtools_amd$lib.so
    MSX_LIB_PATH=$L timeout 300 python scripts/bench_inflate.py /tmp/$f.bam 8192 --skip-bytes 3000000 2>&1 | grep "^lanes" | sed "s/^/[$lib] $f /" | tee -a $OUT/log.txt
  done
done
