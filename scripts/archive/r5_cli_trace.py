#!/usr/bin/env python3
"""Reads the rocprofv3 kernel and memory-copy traces of one command-line run and prints (a) a 30 ms timeline in the
steady state, (b) per inflate launch: the gap to the previous inflate launch's end and what ran on the device in between."""
import csv
import glob
import sys

out = sys.argv[1]
ev = []
for f in glob.glob(out + "/cli/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", "q" + r["Queue_Id"], r["Kernel_Name"][:28]))
for f in glob.glob(out + "/cli/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "", r["Direction"][-24:]))
ev.sort()
inf = [e for e in ev if "k_bgzf_inflate" in e[4]]
print(len(ev), "events,", len(inf), "inflate launches")
if len(inf) < 8:
    sys.exit(0)
mid = inf[len(inf) // 2][0]
with open(out + "/timeline.txt", "w") as fh:
    for s, e, k, q, n in ev:
        if mid - 2e6 <= s <= mid + 28e6:
            fh.write(f"{(s - mid) / 1e6:10.3f} {(e - s) / 1e6:8.3f}  {k} {q} {n}\n")
with open(out + "/per_batch.txt", "w") as fh:
    fh.write("# inflate launch: start (ms), duration, gap since the previous inflate ended, busy time of other work inside that gap\n")
    for i in range(1, len(inf)):
        g0, g1 = inf[i - 1][1], inf[i][0]
        busy = sum(max(0, min(e, g1) - max(s, g0)) for s, e, k, q, n in ev if "inflate" not in n and e > g0 and s < g1 and k == "K")
        cp = sum(max(0, min(e, g1) - max(s, g0)) for s, e, k, q, n in ev if k == "C" and e > g0 and s < g1)
        fh.write(f"{(inf[i][0] - inf[0][0]) / 1e6:9.3f} {(inf[i][1] - inf[i][0]) / 1e6:7.3f} {(g1 - g0) / 1e6:7.3f} kernels {busy / 1e6:7.3f} copies {cp / 1e6:7.3f}\n")
print(open(out + "/per_batch.txt").read()[:3000])
