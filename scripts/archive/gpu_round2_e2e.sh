#!/bin/bash
# round 2: command-line pipeline: parity of the CLI tests, then end-to-end timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/e2e
rm -rf $OUT; mkdir -p $OUT
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $OUT/host.txt 2>&1; cat $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 1500 python scripts/e2e_cli.py ${1:-10000000} 100000 > $OUT/e2e.json 2> $OUT/e2e.err
tail -c 400 $OUT/e2e.err
python3 - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/e2e/e2e.json"))
    print({k:v for k,v in d.items() if k!="runs"})
    for r in d["runs"]: print(r["cmd"], r["s"], r["M_alignments_per_s"], r.get("stages"))
except Exception as e: print("e2e failed", e)
PY
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 - $OUT/bench_c3.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches"]) for k,v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
