#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -8
