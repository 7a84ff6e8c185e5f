#!/bin/bash
# round 3: c2 bench, kernel trace + PMC traffic of the c3 step (the default bench line itself: scripts/gpu_r3_d.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3f
rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-coverage > $OUT/bench_c2.json 2> $OUT/bench_c2.err
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-200
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $ARGS > $OUT/d.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
# the distributed step over a one-rank communicator, kernel trace
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg --force-dist"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dist -- python3 $ARGS > $OUT/trace_dist.log 2>&1
find $OUT/trace_dist -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_dist.csv
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for name in "acd":
    files = glob.glob(f"gpurun_out/r3f/{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(name, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = {"avg": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("gpurun_out/r3f/summary.json", "w"), indent=1)
for k in ("k_share_reduce", "k_aln_stats_flat<false>", "k_besthit_select<true>"):
    print(k, {c: round(v["avg"]) for c, v in res.get(k, {}).items()})
PY
