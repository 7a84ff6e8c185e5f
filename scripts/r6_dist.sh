#!/bin/bash
# round 6: MSX_DIST_SLICES and the integer merge of --multi equal -- their tests, then the one-rank distributed step both ways
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_dist}; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_emulated_ranks.py tests/test_gpu_determinism.py -x -q -m gpu --timeout=600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-coverage > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json; d=json.loads([x for x in open('$OUT/bench.json') if x.startswith('{')][-1]); print('ms_per_step', d['ms_per_step']); print(json.dumps(d.get('dist_one_rank'))[:900])"
