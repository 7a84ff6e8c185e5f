#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_cli_scale.py tests/test_gpu_chains.py tests/test_gpu_unpack.py -m gpu -q -x > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
tail -40 $OUT/new_tests.log
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
for i in 1 2; do
( time MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep -v "^#     Prop" | tail -12
done
( time MSX_TIMING=1 MSX_HOST_UNPACK=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep -v "^#     Prop" | tail -12
( time MSX_TIMING=1 $B profile --label S -o /tmp/p1.gz /tmp/in.bam ) 2>&1 | grep -v "^#     Prop" | tail -8
