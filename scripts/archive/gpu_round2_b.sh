#!/bin/bash
# round 2, second GPU call: the restructured sharing iteration (parity, timings at 3 / 4 waves per SIMD)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
for w in 3 4; do
  MSX_SR_WPS=$w timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_wps$w.json 2> $OUT/bench_wps$w.err
  python3 - $OUT/bench_wps$w.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches"],v["algorithmic_GBps"]) for k,v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 1200 $OUT/bench_full.json
python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
