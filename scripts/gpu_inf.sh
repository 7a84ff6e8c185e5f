#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== full collection, one_rank only"; timeout 900 python -m pytest tests -q -m gpu -k "one_rank" 2>&1 | grep "passed\|failed" | tail -1
echo "== inflate + scale + two_ranks"; timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu 2>&1 | grep "passed\|failed" | tail -1
echo "== parity + scale + two_ranks"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu 2>&1 | grep "passed\|failed" | tail -1
echo "== scale + two_ranks"; timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu 2>&1 | grep "passed\|failed" | tail -1
python -c "
import torch, ctypes
print(torch.cuda.device_count())
import subprocess, os
print([l.split()[-1] for l in open('/proc/self/maps') if 'hsa-runtime' in l or 'amdhip64' in l][:4])
"
