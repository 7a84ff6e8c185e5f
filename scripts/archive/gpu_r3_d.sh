#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3d
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench_default.json').read().strip().split("\n")[-1])
print(d['value'], d['ms_per_step'], d.get('dist_one_rank_ms_per_step'))
e=d['e2e']; print({k:e.get(k) for k in ('M_alignments_per_s','seconds','parity_ok')}); print(e.get('one_process_tee')); print(e.get('parity',{}).get('tee_profile'), e.get('parity',{}).get('tee_filter_ok'))
print(d.get('e2e_seq',{}).get('one_process_tee')); print(e.get('inflate')); print(d.get('e2e_seq',{}).get('inflate')); print({k: e.get(k) for k in ('filter_alone','profile_alone')}); print(d.get('coverage',{}).get('cli'))
PY
cat $OUT/bench_time.txt
