#!/bin/bash
# round 2: parity + bench + PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs) + kernel trace stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 - $OUT/bench_c3.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["value"], d["ms_per_step"], d.get("parity",{}).get("ok"), {k:(v["ms_per_step"],v["launches"],v["algorithmic_GBps"]) for k,v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
timeout 600 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $ARGS > $OUT/d.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/e -- python3 $ARGS > $OUT/e.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for name in "acde":
    files = glob.glob(f"gpurun_out/r2c/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = {"avg": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("gpurun_out/r2c/summary.json", "w"), indent=1)
for k in sorted(res):
    if k.startswith("k_aln") or k.startswith("k_share") or k.startswith("k_besthit") or k.startswith("k_prop_apply"):
        print(k, {c: round(v["avg"] / 1e6, 3) for c, v in res[k].items()})
PY
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
