#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cov
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_host_cli.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "coverage or c4" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python scripts/bench_coverage.py > $OUT/cov_binned.json 2> $OUT/cov_binned.err; tail -1 $OUT/cov_binned.json
MSX_COV_BINNED_FROM=1000000000 python scripts/bench_coverage.py > $OUT/cov_atomics.json 2> $OUT/cov_atomics.err; tail -1 $OUT/cov_atomics.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_coverage.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 scripts/bench_coverage.py > $OUT/c.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 scripts/bench_coverage.py > $OUT/d.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for name in "cd":
    files = glob.glob(f"gpurun_out/cov/{name}/**/*counter_collection.csv", recursive=True)
    if not files: print(name, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        agg[row["Kernel_Name"].split("(")[0].replace("void ","")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items(): res[k][c] = {"avg": sum(v)/len(v), "launches": len(v)}
json.dump(res, open("gpurun_out/cov/summary.json","w"), indent=1)
for k in res:
    if "cov" in k: print(k, {c: round(v["avg"]/1e3,1) for c,v in res[k].items()}, "MB")
PY
