#!/bin/bash
# k_bgzf_inflate / k_bgzf_crc: kernel durations (rocprofv3 --kernel-trace --stats), SQ instruction counters, the wave sweep,
# and the command line's device timeline with inflate ahead
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3inf; rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
ARGS="scripts/bench_inflate.py /tmp/in.bam 8192"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python3 $ARGS > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
for w in 4 8 10 12 14; do echo "waves_per_cu $w $(MSX_INFLATE_WAVES=$w python3 $ARGS 2>&1 | tail -1)"; done > $OUT/wave_sweep.txt
MSX_INFLATE_STATS=1 python3 $ARGS 2>&1 | tail -2 > $OUT/symbols.txt
export MSX_CLEAN_EXIT=1 MSX_TIMING=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/cli -o cli -- $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> $OUT/cli_timing.txt
python3 - <<'PY'
import csv, glob, collections, json
OUT = "gpurun_out/r3inf"
res = {}
for f in glob.glob(OUT + "/stats/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in csv.DictReader(open(f)) if "bgzf" in r["Name"]]
cnt = {}
for d in "ab":
    for f in glob.glob(OUT + f"/{d}/**/*counter_collection.csv", recursive=True):
        acc, calls = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_bgzf_inflate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
        for k, v in acc.items():
            cnt[k] = round(v / max(1, calls[k]))
res["sq_counters_per_launch_8192_blocks"] = cnt
res["wave_sweep"] = open(OUT + "/wave_sweep.txt").read().strip().split("\n")
res["symbols"] = open(OUT + "/symbols.txt").read().strip().split("\n")
ev = []
for f in glob.glob(OUT + "/cli/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r["Queue_Id"], r["Kernel_Name"][:28])))
for f in glob.glob(OUT + "/cli/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s" % r["Direction"][:24]))
ev.sort()
inf = [(s, e) for s, e, n in ev if "k_bgzf_inflate" in n]
res["cli_inflate_ms"] = [round((e - s) / 1e6, 2) for s, e in inf]
res["cli_inflate_period_ms"] = [round((inf[i + 1][0] - inf[i][0]) / 1e6, 2) for i in range(len(inf) - 1)]
mid = inf[len(inf) // 2][0]
with open(OUT + "/timeline_30ms.txt", "w") as out:
    for s, e, n in ev:
        if mid - 2_000_000 <= s < mid + 28_000_000:
            out.write("%10.3f %8.3f  %s\n" % ((s - mid) / 1e6, (e - s) / 1e6, n))
json.dump(res, open(OUT + "/inflate_summary.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
rm -rf $OUT/a $OUT/b $OUT/stats $OUT/cli
