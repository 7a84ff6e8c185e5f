#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3h
rm -rf $OUT; mkdir -p $OUT
export MSX_VALIDATION_OUT=$GRAFT_REPO_ROOT/$OUT/validation_grid.json
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
