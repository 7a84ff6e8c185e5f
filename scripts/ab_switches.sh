#!/bin/bash
# the command-line tests at scale under the A/B switches that select round 3's paths (every one must stay green; MSX_HOST_INFLATE=1
# is left out: tests/test_cli_scale.py::test_where_the_blocks_are_inflated_changes_nothing sets the inflater's switches itself)
cd "$GRAFT_REPO_ROOT"
for e in MSX_DEFLATE_SYNC=1 MSX_HOST_FRAME=1 MSX_HOST_DEFLATE=1 MSX_NO_WARMUP=1 MSX_WRITE_THREADS=4; do
  echo "== $e: $(env $e timeout 900 python -m pytest -m gpu -x -q tests/test_cli_scale.py tests/test_gpu_chains.py 2>&1 | tail -1)"
done
