#!/bin/bash
# PMC passes for the roofline `traffic` field and stall analysis (run on the GPU box).
# Counters are collected in their own runs, with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ("fetch", "write", "sq"):
    files = glob.glob(f"gpurun_out/pmc/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", name)
    for k, cs in sorted(agg.items()):
        if not k.startswith("k_") and "scan" not in k: continue
        print(k, {c: (round(sum(v)/len(v), 1), len(v)) for c, v in cs.items()})
PY
