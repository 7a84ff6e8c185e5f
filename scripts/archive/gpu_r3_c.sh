#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_cli_scale.py tests/test_gpu_chains.py tests/test_host_cli.py -m gpu -q -x > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
tail -40 $OUT/new_tests.log
