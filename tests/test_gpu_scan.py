"""The prefix sum every compaction of the path goes through (msx_scan.hip: reduce-then-scan, up to three levels of 2048-item
chunks) against numpy, by itself, at the sizes where its levels have edges: no item, one chunk, one item more than a chunk,
64 chunks, 2048 chunks and a few items (a third level), 3 x 10^7 items, a device-side length shorter than the launch,
inclusive in place; every call on one context (the levels' workspaces are reused and grown).
The entry point exists in the debug build of the library only (msamtools_amd/dbg): the test runs in a child process that loads it."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
DBG = os.path.join(ROOT, "msamtools_amd", "dbg", "libmsamtools_amd.so")

CHILD = r"""
import ctypes as C, sys
import numpy as np
import msamtools_amd as m
ctx = m.Context(0)
L = ctx.lib
L.msx_debug_scan_u32.restype = C.c_int
L.msx_debug_scan_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
rng = np.random.default_rng(11)
sizes = [0, 1, 7, 2047, 2048, 2049, 4096, 64 * 2048 - 1, 64 * 2048, 64 * 2048 + 1, 65 * 2048 + 5, 1_000_003, 2048 * 2048 + 17, 30_000_001]
cap = max(sizes) + 8
d_in, d_out, d_n = ctx.alloc(4 * cap), ctx.alloc(4 * cap + 8), ctx.alloc(8)
checked = 0
for rep in range(2):
    for n in sizes:
        x = rng.integers(0, 100, n, dtype=np.uint32)
        if n:
            ctx.to_dev(d_in, x)
        # exclusive, out[n] = total
        ctx.check(L.msx_debug_scan_u32(ctx.h, d_in, d_out, n, 0, None))
        got = ctx.to_host(d_out, n + 1, np.uint32)
        want = np.concatenate([[0], np.cumsum(x, dtype=np.uint64)]).astype(np.uint32)
        assert np.array_equal(got, want), ("exclusive", n)
        # inclusive, in place
        if n:
            ctx.to_dev(d_out, x)
            ctx.check(L.msx_debug_scan_u32(ctx.h, d_out, d_out, n, 1, None))
            got = ctx.to_host(d_out, n, np.uint32)
            assert np.array_equal(got, want[1:]), ("inclusive", n)
        # a device-side length: the items at and beyond it count as zeros and are not written; out[n] is the total
        if n > 10:
            k = int(rng.integers(1, n))
            ctx.to_dev(d_n, np.array([k], np.uint64))
            ctx.to_dev(d_out, np.full(n + 1, 0xdeadbeef, np.uint32))
            ctx.check(L.msx_debug_scan_u32(ctx.h, d_in, d_out, n, 0, d_n))
            got = ctx.to_host(d_out, n + 1, np.uint32)
            assert np.array_equal(got[:k], want[:k]), ("device-side length", n, k)
            assert got[n] == want[k], ("device-side length: total", n, k)
        checked += 1
print("checked", checked)
"""


def test_scan_against_numpy_at_the_edges_of_its_chain():
    assert os.path.exists(DBG), "make -C msamtools_amd/csrc builds the debug library"
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=dict(os.environ, MSX_LIB_PATH=DBG, PYTHONPATH=ROOT),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert b"checked 28" in r.stdout
