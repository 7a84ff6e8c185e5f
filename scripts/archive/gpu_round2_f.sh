#!/bin/bash
# round 2: full default bench line (with cpu_baseline, parity, e2e) + c2 + kernel trace + PMC traffic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2f
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python3 - $OUT/bench_default.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("parity",{}).get("ok"))
    print({k:(v["ms_per_step"],v["launches"],v["algorithmic_GBps"]) for k,v in d["roofline"]["per_kernel"].items()})
    print("cpu", d["cpu_baseline"]["value"], d.get("cpu_baseline_all_cores",{}).get("value"))
    print("e2e", json.dumps(d.get("e2e"))[:1500])
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
sleep 5   # (right after the command-line runs of the default bench the device is still releasing their memory: c2 measured 25 % slower there)
timeout 600 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-200
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $ARGS > $OUT/d.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/e -- python3 $ARGS > $OUT/e.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for name in "acde":
    files = glob.glob(f"gpurun_out/r2f/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = {"avg": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("gpurun_out/r2f/summary.json", "w"), indent=1)
PY
