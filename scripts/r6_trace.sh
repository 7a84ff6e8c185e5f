#!/bin/bash
# round 6: where the c3 step's time goes on the timeline -- kernel trace of three steps, gaps between the main chain's kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_trace}; rm -rf $OUT; mkdir -p $OUT
STEP="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg ${2:-}"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $STEP > $OUT/run.log 2>&1
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ks.sort()
# the last step: from the last k_aln_stats_flat on
i0 = max(i for i, k in enumerate(ks) if "k_aln_stats_flat" in k[2])
step = ks[i0:]
t0 = step[0][0]
end = max(k[1] for k in step if "k_prop_purged" in k[2]) if any("k_prop_purged" in k[2] for k in step) else step[-1][1]
print(f"step: {(end - t0) / 1e3:.1f} us from the start of k_aln_stats_flat to the end of k_prop_purged; {len(step)} kernels")
busy, last_end, gaps = 0, t0, 0
# union of busy intervals (any queue)
iv = sorted((a, b) for a, b, _, _ in step if a < end)
cur_a, cur_b = iv[0]
for a, b in iv[1:]:
    if a > cur_b:
        busy += cur_b - cur_a; gaps += a - cur_b; cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
print(f"device busy (any queue) {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us")
prev_end = {}
for a, b, n, q in step:
    if a >= end: break
    g = a - prev_end.get(q, a)
    print(f"{(a - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f} us  gap {g / 1e3:6.1f}  q{q}  {n}")
    prev_end[q] = b
PY
head -3 $OUT/timeline.txt; tail -3 $OUT/run.log | cut -c1-300
