#!/bin/bash
# L2 / fabric counters of the sharing-iteration kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $ARGS > $OUT/d.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/e -- python3 $ARGS > $OUT/e.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for name in "abcde":
    files = glob.glob(f"gpurun_out/pmc_l2/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file", open(f"gpurun_out/pmc_l2/{name}.log").read()[-400:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = {"avg": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("gpurun_out/pmc_l2/summary.json", "w"), indent=1)
for k in sorted(res):
    if k.startswith("k_") or "scan" in k:
        print(k, {c: round(v["avg"] / 1e6, 3) for c, v in res[k].items()}, "(millions, avg per launch)")
PY
