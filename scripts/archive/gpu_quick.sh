#!/bin/bash
# quick GPU check: parity tests + the c3 bench line (+ optional extra env runs given as arguments "NAME=VALUE")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
run_bench() {
  timeout 900 env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python3 - $OUT/bench_$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches"],v["algorithmic_GBps"]) for k,v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
run_bench X=1
for e in "$@"; do run_bench "$e"; done
