#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/dbg; rm -rf $OUT; mkdir -p $OUT
B=msamtools_amd/bin/msamtools
$B synth --groups 60000 --refs 500 -b > /tmp/in.bam
$B profile --label S --multi prop -o /tmp/a.gz /tmp/in.bam 2> $OUT/a.err
MSX_FORCE_DIST=1 $B profile --label S --multi prop -o /tmp/b.gz /tmp/in.bam 2> $OUT/b.err
MSX_FORCE_DIST=1 MSX_SERIAL_IO=1 $B profile --label S --multi prop -o /tmp/c.gz /tmp/in.bam 2> $OUT/c.err
MSX_FORCE_DIST=1 MSX_SERIAL=1 $B profile --label S --multi prop -o /tmp/d.gz /tmp/in.bam 2> $OUT/d.err
for x in a b c d; do echo "== $x"; zcat /tmp/$x.gz | head -16 | tail -12; tail -8 $OUT/$x.err; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
