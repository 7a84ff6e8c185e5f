"""The floating-point outputs repeat bit for bit (-m gpu): the same inserts through the same build give the same abundance
BITS run after run, on the same context and on a fresh one -- every sum of the sharing iteration has a fixed order (segments
summed lane by lane, partial slots added in slot order, diff^2 in workgroup order: DESIGN.md section 3; no floating-point
atomics anywhere on the path).  A race or an order that depends on scheduling would still pass the 1e-6 parity tests; it
shows here.  --multi equal (its d[] is accumulated with integer-exact 1/k additions per pool in pool order) likewise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OPTS = dict(l=80, p=95, z=80, besthit=True)


def one_run(m, ctx, seed, groups, refs, multi, fused, dist=False):
    db = m.DeviceBatch.synth(ctx, seed, groups, refs, 4)
    prof = m.Profile(ctx, refs, multi)
    if fused:
        run = m.FilterRun(ctx, db, **OPTS)
        run.enqueue_with_profile(prof)
        run.finish()
        run.free()
    else:
        prof.accumulate(db, None)
    if dist:
        prof.finalize_dist_enqueue()
    else:
        prof.finalize_enqueue()
    ab, st = prof.fetch()
    out = (ab.copy(), [st.delta[k] for k in range(20)], st.iterations)
    prof.close()
    db.free()
    return out


@pytest.mark.parametrize("groups,refs", [(3000, 40), (250_000, 3000), (1_200_000, 150_000)])
@pytest.mark.parametrize("multi", ["proportional", "equal"])
@pytest.mark.parametrize("fused", [True, False])
def test_abundance_bits_repeat(groups, refs, multi, fused):
    import msamtools_amd as m
    ctx = m.Context(0)
    try:
        first = one_run(m, ctx, 13579, groups, refs, multi, fused)
        assert first[0].sum() > 0
        for _ in range(4):
            # (another shape in between: scratch is regrown and reused)
            one_run(m, ctx, 5, 7000, 90, multi, fused)
            again = one_run(m, ctx, 13579, groups, refs, multi, fused)
            assert np.array_equal(first[0].view(np.uint64), again[0].view(np.uint64))
            assert first[1] == again[1] and first[2] == again[2]             # DELTA^2 of every iteration, bit for bit
    finally:
        ctx.close()
    ctx2 = m.Context(0)
    try:
        fresh = one_run(m, ctx2, 13579, groups, refs, multi, fused)
    finally:
        ctx2.close()
    assert np.array_equal(first[0].view(np.uint64), fresh[0].view(np.uint64)) and first[1] == fresh[1]


@pytest.mark.parametrize("groups,refs", [(250_000, 3000), (1_200_000, 150_000)])
def test_side_lanes_change_nothing(groups, refs):
    """msx_ctx_set_lanes: a context's side lanes (streams on which a step's independent scans and compactions overlap; the command
    line turns them off, the Python layer and bench.py leave them on) decide WHEN kernels run, not what they compute: the same
    abundance bits, the same DELTA^2 of every iteration, the same emit list with and without them."""
    import msamtools_amd as m
    res = []
    for lanes in (1, 0, 1):
        ctx = m.Context(0)
        try:
            ctx.check(ctx.lib.msx_ctx_set_lanes(ctx.h, lanes))
            db = m.DeviceBatch.synth(ctx, 24680, groups, refs, 4)
            prof = m.Profile(ctx, refs, "proportional")
            run = m.FilterRun(ctx, db, **OPTS)
            run.enqueue_with_profile(prof)
            run.finish()
            emit = run.result().emit.copy()
            run.free()
            prof.finalize_enqueue()
            ab, st = prof.fetch()
            res.append((ab.copy(), [st.delta[k] for k in range(20)], st.iterations, emit))
            prof.close()
            db.free()
        finally:
            ctx.close()
    for other in res[1:]:
        assert np.array_equal(res[0][0].view(np.uint64), other[0].view(np.uint64))
        assert res[0][1] == other[1] and res[0][2] == other[2]
        assert np.array_equal(res[0][3], other[3])
