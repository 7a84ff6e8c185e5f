#!/usr/bin/env python3
"""msx_bgzf_inflate on the blocks of a BAM file: rate, and the output against zlib."""
import ctypes as C
import struct
import sys
import time
import zlib

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import msamtools_amd as m
from msamtools_amd import _lib as L

path, nblk = sys.argv[1], int(sys.argv[2])
raw = open(path, "rb").read(nblk * 70000)
blocks, pos = [], 0
while pos + 18 <= len(raw) and len(blocks) < nblk:
    xlen = struct.unpack_from("<H", raw, pos + 10)[0]
    bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
    if pos + bsize > len(raw):
        break
    crc, isize = struct.unpack_from("<II", raw, pos + bsize - 8)
    blocks.append((pos + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
    pos += bsize
n = len(blocks)
arr = (L.BgzfBlock * n)()
uo = 0
for i, (io, il, ol, crc) in enumerate(blocks):
    arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_len, arr[i].crc32 = io, il, uo, ol, crc
    uo += ol
ctx = m.Context(0)
comp = np.frombuffer(raw[:pos], np.uint8)
d_comp, d_blk, d_out, d_st = ctx.alloc(pos + 64), ctx.alloc(32 * n), ctx.alloc(uo + 64), ctx.alloc(4 * n)
ctx.to_dev(d_comp, comp)
ctx.to_dev(d_blk, np.frombuffer(bytes(arr), np.uint8))
ref = C.c_int64()
ts = []
for it in range(6):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.msx_bgzf_inflate(ctx.h, C.c_void_p(d_comp), pos, C.c_void_p(d_blk), n, C.c_void_p(d_out), C.c_void_p(d_st), C.byref(ref)))
    ts.append(time.perf_counter() - t0)
best = min(ts[1:])
out = ctx.to_host(d_out, uo, np.uint8).tobytes()
st = ctx.to_host(d_st, n, np.uint32)
ok = True
o = 0
t0 = time.perf_counter()
for io, il, ol, crc in blocks[:2000]:
    if zlib.decompress(raw[io:io + il], -15) != out[o:o + ol]:
        ok = False
        break
    o += ol
print(f"{n} blocks, {pos / 1e6:.1f} MB -> {uo / 1e6:.1f} MB: {best * 1e3:.3f} ms = {uo / best / 1e9:.1f} GB/s out; refused {ref.value}, status!=0: {int((st != 0).sum())}, first 2000 blocks equal zlib: {ok}")
