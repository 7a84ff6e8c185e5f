#!/usr/bin/env python3
"""BASELINE.json configs[3]: per-base coverage on 50 k refs x 5 kb, 50 M alignments, 1 GPU.
Prints one JSON line (not the driver's bench contract; see bench.py for that)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import msamtools_amd as m

ngrp = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
refs, tl = 50_000, 5_000
ctx = m.Context(0)
db = m.DeviceBatch.synth(ctx, 13579, ngrp, refs, 4)
off = np.arange(refs + 1, dtype=np.int64) * tl
total = int(off[-1])
d_off = ctx.alloc(off.nbytes)
d_cov = ctx.alloc(4 * total + 8)
ctx.to_dev(d_off, off)
ctx.timing(True)
ts = []
for it in range(6):
    ctx.zero(d_cov, 4 * total + 8)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.msx_coverage_accumulate(ctx.h, C.byref(db.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
    ctx.check(ctx.lib.msx_coverage_finish(ctx.h, C.c_void_p(d_cov), total))
    ctx.sync()
    ts.append(time.perf_counter() - t0)
pile_ms, pile_n = ctx.timing_get("k_coverage_pileup")
scan_ms, scan_n = ctx.timing_get("scan")
cov = ctx.to_host(d_cov, total, np.int32)
best = min(ts[1:])
# CPU oracle on a sample
import oracle_lib as orc
hs = m.HostSynth(13579, 200_000, refs, 4)
t0 = time.perf_counter()
orc.coverage(hs, [tl] * refs)
cpu = time.perf_counter() - t0
# algorithmic bytes of the pile-up: tid 4, pos 4, cigar_off 4, cigar words in; two 4-byte marks per run out (read-modify-write)
sz = db.sizes
runs = db.n_records          # the synthetic records have one M/=/X run each, a deletion splits 2.5 % of them in two
pile_bytes = 12 * db.n_records + 4 * int(sz.n_cigar) + 16 * runs
print(json.dumps({
    "workload": f"coverage: {db.n_records} alignments, {refs} refs x {tl} bp ({4 * total / 1e9:.2f} GB of int32 depths)",
    "M_alignments_per_s": round(db.n_records / best / 1e6, 1), "ms": round(best * 1e3, 3),
    "k_coverage_pileup_ms": round(pile_ms / pile_n, 3), "prefix_sum_ms": round(scan_ms / scan_n, 3),
    "path": "binned" if db.n_records >= int(os.environ.get("MSX_COV_BINNED_FROM", 2 << 20)) else "atomics",
    "pileup_algorithmic_GBps": round(pile_bytes / (pile_ms / pile_n * 1e-3) / 1e9, 1),
    "pileup_marks_per_s_G": round(2 * runs / (pile_ms / pile_n * 1e-3) / 1e9, 2),
    "prefix_sum_GBps": round(2 * 4 * total / (scan_ms / scan_n * 1e-3) / 1e9, 1),
    "depth_sum": int(cov.astype(np.int64).sum()),
    "cpu_oracle_M_alignments_per_s": round(hs.n_records / cpu / 1e6, 2)}))
