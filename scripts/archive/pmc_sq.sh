#!/bin/bash
# SQ counters for stall analysis of one or all kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ("a", "b"):
    files = glob.glob(f"gpurun_out/pmc_sq/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file", open(f"gpurun_out/pmc_sq/{name}.log").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in ("k_aln_stats_filter", "k_besthit_select", "k_insert_count", "k_share_reduce", "k_list_recip"):
        if k in agg: print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in agg[k].items()}, "(millions)")
PY
