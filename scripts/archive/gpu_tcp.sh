#!/bin/bash
# vector-cache / texture-addresser counters of one kernel family (argument: substring of the kernel name), c3, one step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-k_share_reduce}
OUT=gpurun_out/tcp
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e"
i=0
# (counter sets with TA_*, TCP_GATE_EN*, TCP_*_STALL_CYCLES or TCP_TOTAL_* made rocprofv3 sit until the timeout on
#  this pool -- 5 GPU-minutes each: only sets seen to work are listed)
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_VOLATILE_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 $ARGS > $OUT/s$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, collections, json, sys
K = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/tcp/s*/")):
    files = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counters:", open(d.rstrip("/") + ".log").read()[-300:].replace("\n", " | ")); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
json.dump(res, open("gpurun_out/tcp/summary.json", "w"), indent=1)
for k in res:
    if K in k: print(k, json.dumps({c: round(v) for c, v in res[k].items()}, indent=0))
PY
