#!/usr/bin/env python3
"""c3: k_besthit_select with the insert accounting inside (filter | profile fused) and without it (filter alone), per
kernel by the library's HIP-event timers.  usage: bench_select.py [groups] [refs]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msamtools_amd as m

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nrefs = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = m.Context(0)
db = m.DeviceBatch.synth(ctx, 13579, ng, nrefs, 4)
run = m.FilterRun(ctx, db, l=80, p=95, z=80, besthit=True)
prof = m.Profile(ctx, nrefs, "proportional")
names = ["k_aln_stats_flat", "k_besthit_select", "k_emit_order", "k_insert_count", "k_multi_compact"]
res = {"records": db.n_records}
for mode in ("fused", "filter_only"):
    for rep in range(2):
        prof.reset()
        run.enqueue_with_profile(prof) if mode == "fused" else run.enqueue()
        run.finish()
    ctx.timing(True)
    ctx.timing_reset()
    reps = 4
    for _ in range(reps):
        prof.reset()
        run.enqueue_with_profile(prof) if mode == "fused" else run.enqueue()
        run.finish()
    res[mode] = {k: round(ctx.timing_get(k)[0] / reps, 4) for k in names}
    ctx.timing(False)
print(json.dumps(res))
