#!/usr/bin/env python3
"""msx_bgzf_inflate on the blocks of a BAM file: rate of the lane-parallel kernel (the default) and of the serial one (MSX_INFLATE_SERIAL=1)
on the same blocks in one process, and EVERY block's output against zlib's.
usage: bench_inflate.py file.bam n_blocks [--skip-bytes N] [--json out.json]"""
import ctypes as C
import json
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msamtools_amd as m
from msamtools_amd import _lib as L

path, nblk = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[sys.argv.index("--skip-bytes") + 1]) if "--skip-bytes" in sys.argv else 0
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
with open(path, "rb") as f:
    f.seek(0)
    raw = f.read(skip + nblk * 70000)
blocks, pos = [], 0
while pos + 18 <= len(raw) and len(blocks) < nblk:
    xlen = struct.unpack_from("<H", raw, pos + 10)[0]
    bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
    if pos + bsize > len(raw):
        break
    crc, isize = struct.unpack_from("<II", raw, pos + bsize - 8)
    if pos >= skip:
        blocks.append((pos + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
    pos += bsize
n = len(blocks)
arr = (L.BgzfBlock * n)()
uo = 0
for i, (io, il, ol, crc) in enumerate(blocks):
    arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_len, arr[i].crc32 = io, il, uo, ol, crc
    uo += ol
cin = sum(b[1] for b in blocks)
ctx = m.Context(0)
comp = np.frombuffer(raw[:pos], np.uint8)
d_comp, d_blk, d_out, d_st = ctx.alloc(pos + 64), ctx.alloc(32 * n), ctx.alloc(uo + 64), ctx.alloc(4 * n)
ctx.to_dev(d_comp, comp)
ctx.to_dev(d_blk, np.frombuffer(bytes(arr), np.uint8))
ref = C.c_int64()
t0 = time.perf_counter()
want = b"".join(zlib.decompress(raw[io:io + il], -15) for io, il, ol, crc in blocks)
zlib_s = time.perf_counter() - t0
res = {"file": os.path.basename(path), "blocks": n, "compressed_MB": round(cin / 1e6, 1), "inflated_MB": round(uo / 1e6, 1),
       "zlib_one_core_GBps": round(uo / zlib_s / 1e9, 3)}
for name, env in (("lanes", None), ("serial", "1")):
    if env:
        os.environ["MSX_INFLATE_SERIAL"] = env
    else:
        os.environ.pop("MSX_INFLATE_SERIAL", None)
    ts = []
    for it in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.msx_bgzf_inflate(ctx.h, C.c_void_p(d_comp), pos, C.c_void_p(d_blk), n, C.c_void_p(d_out), C.c_void_p(d_st), C.byref(ref)))
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    out = ctx.to_host(d_out, uo, np.uint8).tobytes()
    st = ctx.to_host(d_st, n, np.uint32)
    equal = out == want
    first_bad = None
    if not equal:
        o = 0
        for bi, (io, il, ol, crc) in enumerate(blocks):
            if out[o:o + ol] != want[o:o + ol]:
                k = next(k for k in range(ol) if out[o + k] != want[o + k])
                first_bad = {"block": bi, "byte": k, "of": ol, "in_len": il}
                break
            o += ol
    res[name] = {"ms": round(best * 1e3, 3), "GBps_out": round(uo / best / 1e9, 1), "refused": int(ref.value),
                 "status_nonzero": int((st != 0).sum()), "every_block_equals_zlib": bool(equal), "first_bad": first_bad}
    print(f"{name:6s}: {n} blocks, {cin / 1e6:.1f} MB -> {uo / 1e6:.1f} MB: {best * 1e3:.3f} ms = {uo / best / 1e9:.1f} GB/s out; "
          f"refused {ref.value}, status!=0: {int((st != 0).sum())}, every block equals zlib: {equal} {first_bad or ''}", flush=True)
    # the memory the next kernel writes must not already hold the answer
    ctx.to_dev(d_out, np.zeros(uo, np.uint8))
os.environ.pop("MSX_INFLATE_SERIAL", None)
if jpath:
    with open(jpath, "w") as f:
        json.dump(res, f, indent=1)
