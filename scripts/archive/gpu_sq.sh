#!/bin/bash
# SQ counters of one kernel family (argument: substring of the kernel name), c3 batch, one step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-k_besthit_select}
OUT=gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, collections, json, sys
K = sys.argv[1]
res = collections.defaultdict(dict)
for name in "abc":
    files = glob.glob(f"gpurun_out/sq/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file", open(f"gpurun_out/sq/{name}.log").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
json.dump(res, open("gpurun_out/sq/summary.json", "w"), indent=1)
for k in res:
    if K in k: print(k, json.dumps({c: round(v) for c, v in res[k].items()}))
PY
