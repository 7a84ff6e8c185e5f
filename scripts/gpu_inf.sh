#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== collect inflate, run one_rank only"; timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu -k "one_rank" 2>&1 | grep "passed\|failed" | tail -1
cat > /tmp/test_aaa.py <<'PY'
import zlib
import numpy as np
import pytest
pytestmark = pytest.mark.gpu
def test_z():
    assert True
PY
echo "== dummy module first"; timeout 900 python -m pytest /tmp/test_aaa.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu -k "one_rank" -p no:cacheprovider 2>&1 | grep "passed\|failed" | tail -1
echo "== two_ranks first on the command line"; timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_scale.py -q -m gpu -k "one_rank" 2>&1 | grep "passed\|failed" | tail -1
echo "== unpack + scale + two_ranks"; timeout 900 python -m pytest tests/test_gpu_unpack.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu -k "one_rank" 2>&1 | grep "passed\|failed" | tail -1
echo "== chains + scale + two_ranks"; timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_scale.py tests/test_gpu_two_ranks.py -q -m gpu -k "one_rank" 2>&1 | grep "passed\|failed" | tail -1
