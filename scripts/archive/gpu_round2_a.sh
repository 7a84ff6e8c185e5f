#!/bin/bash
# round 2, first GPU call: parity of the new statistics kernel, A/B against the round-1 kernel, SQ counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
nproc > $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; free -g >> $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_fullsize.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q > $OUT/pytest_full.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_full.log
tail -5 $OUT/pytest_full.log
MSX_STATS_V1=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_v1.json 2> $OUT/bench_v1.err
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench_flat.json 2> $OUT/bench_flat.err
tail -c 600 $OUT/bench_flat.json
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq_a -- python3 $ARGS > $OUT/sq_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_b -- python3 $ARGS > $OUT/sq_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ("sq_a", "sq_b"):
    files = glob.glob(f"gpurun_out/r2a/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in agg:
        if "stats" in k or "besthit" in k:
            print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in agg[k].items()}, "(millions)")
PY
