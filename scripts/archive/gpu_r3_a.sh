#!/bin/bash
# round 3, first GPU call: the new parity tests first (chains, command line at scale), then the whole gpu suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_cli_scale.py -m gpu -q -x > $OUT/new_tests.log 2>&1; echo "rc=$?" >> $OUT/new_tests.log
tail -30 $OUT/new_tests.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
