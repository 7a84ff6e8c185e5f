#!/bin/bash
# kernel trace of the c3 bench (3 steps): per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/trace.log 2>&1
find $OUT/t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/t -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/trace/kernel_stats.csv")))
for r in rows[:34]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"])
PY
tail -2 $OUT/trace.log | cut -c1-300
