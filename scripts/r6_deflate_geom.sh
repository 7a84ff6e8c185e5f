#!/bin/bash
# round 6: the encoder's geometries with 16-bit table entries (debug library: MSX_DEFLATE_GEOM); inputs large enough for several
# rounds of waves (a launch of fewer blocks than waves measures one block's latency, not the rate)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev
export MSX_LIB_PATH=$GRAFT_REPO_ROOT/msamtools_amd/dbg/libmsamtools_amd.so
$D synth --groups 3200000 --refs 1000 -u > /tmp/dl.bam
$D synth --groups 1200000 --refs 1000 --seq -u > /tmp/ds.bam
for f in /tmp/dl.bam /tmp/ds.bam; do
  for g in 0 1 2 3 4 5; do
    echo "geom $g $(basename $f): $(MSX_DEFLATE_STATS=1 MSX_DEFLATE_GEOM=$g python3 scripts/bench_deflate.py $f 2>&1 | grep -E 'level 6|waves per' | sort -u | cut -c1-140 | tr '\n' ' ')"
  done
done
