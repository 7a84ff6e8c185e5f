#!/bin/bash
# kernel experiments: parity of the profile path, then the c3 step with per-kernel times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3k
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_prop_edges.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
run_bench() {
  name=$1; shift
  timeout 900 env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-coverage --no-dist-leg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python3 - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    r=d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "merged", r["merged_lists"], r["merged_entries"], {k:(v["ms_per_step"],v["launches"]) for k,v in r["per_kernel"].items() if v["ms_per_step"]>0.02})
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
run_bench base X=1
for e in "$@"; do run_bench "$(echo $e | tr '=,' '__')" $(echo $e | tr ',' ' '); done
