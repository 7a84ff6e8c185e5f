#!/usr/bin/env python3
"""k_bgzf_deflate alone on the records of a synthetic BAM (MSX_DEFLATE_STATS=1 prints the clocks per phase and block).
usage: bench_deflate.py FILE.bam        -- a BAM written beforehand (`msamtools-dev synth ... -u > FILE.bam`); no child process
       bench_deflate.py [groups] [--seq] -- synthesises the file itself through a child process: NOT under rocprofv3 (the
                                           profiler's preloaded library has initialised the GPU in every child, and a child
                                           that execs is what the pool forbids) -- the script refuses that combination."""
import gzip
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import numpy as np

import msamtools_amd as m

groups = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200000
seq = "--seq" in sys.argv
path = sys.argv[1] if len(sys.argv) > 1 and os.path.isfile(sys.argv[1]) else None
if path:
    raw = gzip.decompress(open(path, "rb").read())
else:
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")):
        sys.exit("bench_deflate.py: under rocprofv3 give it a file (msamtools-dev synth ... -u > f.bam); it starts no child there")
    exe = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
    args = [exe, "synth", "--groups", str(groups), "--refs", "1000", "-u"] + (["--seq"] if seq else [])
    raw = gzip.decompress(subprocess.run(args, stdout=subprocess.PIPE, check=True).stdout)
data = raw[len(raw) // 8:]
data = data[:len(data) // 0xff00 * 0xff00]
ctx = m.Context(0)
n = len(data)
cap = int(ctx.lib.msx_bgzf_bound(n, 6)) + 64
d_in, d_out = ctx.alloc(n + 64), ctx.alloc(cap)
ctx.to_dev(d_in, np.frombuffer(data + b"\0" * 64, np.uint8))
for level in (0, 6):
    best = 1e9
    for rep in range(4):
        n_out, n_blk = C.c_int64(0), C.c_int64(0)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.msx_bgzf_deflate(ctx.h, C.c_void_p(d_in), n, level, C.c_void_p(d_out), cap, C.byref(n_out), C.byref(n_blk)))
        best = min(best, time.perf_counter() - t0)
    z = 0
    if level:
        for i in range(0, min(n, 40 * 0xff00), 0xff00):
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            z += len(co.compress(data[i:i + 0xff00]) + co.flush()) + 26
        z = z * (n / min(n, 40 * 0xff00))
    print(f"level {level}: {n / 1e6:.1f} MB in {n_blk.value} blocks -> {n_out.value / 1e6:.1f} MB in {best * 1e3:.2f} ms = {n / best / 1e9:.1f} GB/s"
          + (f"; zlib -6 would write {z / 1e6:.1f} MB ({n_out.value / z:.3f} x)" if level else ""))
