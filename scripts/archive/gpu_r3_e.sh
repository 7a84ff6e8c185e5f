#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_cli_scale.py tests/test_gpu_chains.py tests/test_gpu_unpack.py tests/test_host_cli.py -m gpu -q -x > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
tail -12 $OUT/new_tests.log
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
for i in 1 2 3; do { time MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p$i.gz --label S /tmp/in.bam > /tmp/f$i.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "filter pipeline" /tmp/err.txt | sed 's/99992794 records.*//' | tr '\n' ' '; cat /tmp/time.txt; sleep 2; done
$B digest /tmp/f3.bam
