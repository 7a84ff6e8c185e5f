#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_unpack.py tests/test_cli_scale.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -15
