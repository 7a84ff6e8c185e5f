#!/usr/bin/env python3
"""BASELINE.json configs[3] alone (for rocprofv3 runs): per-base coverage on 50 k refs x 5 kb, ~50 M alignments, one GPU,
msx_coverage_depths (the whole-sample form) and the streamed form (zero + accumulate + finish).
usage: bench_coverage.py [groups] [reps] [depths|streamed|both]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msamtools_amd as m

ngrp = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
which = sys.argv[3] if len(sys.argv) > 3 else "both"
refs, tl = 50_000, 5_000
ctx = m.Context(0)
db = m.DeviceBatch.synth(ctx, 13579, ngrp, refs, 4)
off = np.arange(refs + 1, dtype=np.int64) * tl
total = int(off[-1])
d_off = ctx.alloc(off.nbytes)
d_cov = ctx.alloc(4 * total + 8)
ctx.to_dev(d_off, off)
res = {"workload": f"coverage: {db.n_records} alignments, {refs} refs x {tl} bp ({4 * total / 1e9:.2f} GB of int32 depths)"}
if which in ("depths", "both"):
    ts = []
    for it in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.msx_coverage_depths(ctx.h, C.byref(db.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
        ts.append(time.perf_counter() - t0)
    res["depths_ms"] = round(min(ts[1:]) * 1e3, 3)
    res["depth_sum"] = int(ctx.to_host(d_cov, total, np.int32).astype(np.int64).sum())
if which in ("streamed", "both"):
    ts = []
    for it in range(reps):
        ctx.zero(d_cov, 4 * total + 8)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.msx_coverage_accumulate(ctx.h, C.byref(db.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
        ctx.check(ctx.lib.msx_coverage_finish(ctx.h, C.c_void_p(d_cov), total))
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    res["streamed_ms"] = round(min(ts[1:]) * 1e3, 3)
    res["depth_sum_streamed"] = int(ctx.to_host(d_cov, total, np.int32).astype(np.int64).sum())
sz = db.sizes
res["algorithmic_bytes"] = 4 * total + 12 * db.n_records + 4 * int(sz.n_cigar)
print(json.dumps(res))
