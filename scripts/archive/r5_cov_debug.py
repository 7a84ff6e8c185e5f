import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import msamtools_amd as m
os.environ["MSX_COV_DEBUG"] = "1"
ctx = m.Context(0)
refs, tl, G = 50000, 5000, 10_000_000
off = np.arange(refs + 1, dtype=np.int64) * tl
total = int(off[-1])
d_off, d_cov = ctx.alloc(off.nbytes), ctx.alloc(4 * total + 8)
ctx.to_dev(d_off, off)
cuts = [0, 1_000_000, 1_000_017, 3_500_000, 3_500_000 + 9_000, 6_000_000, 9_999_999, G]
for lo, hi in zip(cuts[:-1], cuts[1:]):
    part = m.DeviceBatch.synth(ctx, 13579, hi - lo, refs, 4, first_group=lo)
    ctx.check(ctx.lib.msx_coverage_collect(ctx.h, C.byref(part.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
    part.free()
n = C.c_int64(-1)
ctx.check(ctx.lib.msx_coverage_collect_finish(ctx.h, C.c_void_p(d_cov), total, C.byref(n)))
ctx.sync()
print("streamed", n.value)
