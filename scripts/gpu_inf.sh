#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k coverage 2>&1 | tail -3
