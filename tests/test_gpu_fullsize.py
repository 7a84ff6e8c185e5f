"""Full-size GPU checks of BASELINE.json configs[2] and configs[3].

c3 (100 M alignments, 1 M references, `filter -l 80 -p 95 -z 80 --besthit |
profile --multi=proportional`): the WHOLE batch is copied back and put through
the CPU oracle (about 7 s), so record selection, output order, per-reference
counts and counters are compared bit for bit and the proportional profile to
1e-6 relative at the size the bench line is quoted on -- plus the
size-independent properties of test_gpu_scale.py and a 200 000-pool batch with
n_refs = 1 M through the separate (non-fused) entry points.
Reference rules: msam_filter.c:206-245, msam_profile.c:331-405.

c4 (50 M alignments on 50 k references x 5 kb, `coverage`): depth_sum ==
sum of the M/=/X run lengths, and every per-base depth equal to the oracle's
(msam_coverage.c:66-78).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

OPTS = dict(l=80, p=95, z=80, besthit=True)
REL_TOL = 1e-6       # north_star tolerance for the floating-point profile
SEED = 13579

C3_GROUPS, C3_REFS = 20_000_000, 1_000_000
C3_PREFIX = 200_000
C4_GROUPS, C4_REFS, C4_TLEN = 10_000_000, 50_000, 5_000


class HostCopy:
    """A device batch copied back to the host, shaped like samio.Records for oracle_lib."""

    def __init__(self, db):
        h = db.to_host()
        for k in ("flag", "rflags", "tid", "pos", "nm", "as_", "cigar_off", "md_off", "cigar", "md", "group_off"):
            setattr(self, k, h[k])
        ng = db.n_groups
        self.name_id = np.repeat(np.arange(ng, dtype=np.int32), np.diff(self.group_off.astype(np.int64)))
        self.qname_off = self.qname = None
        self.n_records, self.n_groups = db.n_records, ng


@pytest.fixture(scope="module")
def c3():
    import msamtools_amd as m
    ctx = m.Context(0)
    db = m.DeviceBatch.synth(ctx, SEED, C3_GROUPS, C3_REFS, 4)
    run = m.FilterRun(ctx, db, **OPTS)
    prof = m.Profile(ctx, C3_REFS, "proportional")
    run.enqueue_with_profile(prof)          # the fused pipe: what bench.py times
    prof.finalize_enqueue()
    run.finish()
    res = run.result()
    ui = prof.ui()
    ab, st = prof.fetch()
    hs = HostCopy(db)
    want_f = orc.run_filter(hs, **OPTS)
    want_p = orc.run_profile(hs, C3_REFS, multi="proportional", sel=want_f["emit"])
    yield dict(ctx=ctx, db=db, run=run, prof=prof, res=res, ui=ui, ab=ab.copy(), st=st, hs=hs, f=want_f, p=want_p)
    prof.close()
    run.free()
    db.free()
    ctx.close()


def test_c3_shape(c3):
    db = c3["db"]
    assert db.n_groups == C3_GROUPS and 95_000_000 < db.n_records < 105_000_000


def test_c3_filter_equals_oracle_whole_batch(c3):
    """Record selection and output order of all ~100 M records, bit for bit (msam_filter.c:206-245)."""
    res, want = c3["res"], c3["f"]
    assert want["rc"] == 0
    assert res.n_emit == len(want["emit"])
    assert np.array_equal(res.emit, want["emit"])
    kept = np.zeros(c3["db"].n_records, bool)
    kept[want["emit"]] = True
    assert np.array_equal(res.keep != 0, kept)


def test_c3_profile_equals_oracle_whole_batch(c3):
    """Counts exact, proportional abundances <= 1e-6 relative, same iteration count (msam_profile.c:331-405)."""
    st, ref = c3["st"], c3["p"]
    s = ref["stats"]
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
    assert np.array_equal(c3["ui"], ref["ui"])
    assert (st.iterations, st.converged) == (s.iterations, s.converged)
    ab, want = c3["ab"], ref["abundance"]
    assert np.array_equal(ab == 0, want == 0)
    rel = np.abs(ab - want) / np.maximum(np.abs(want), 1e-300)
    assert rel.max() <= REL_TOL, rel.max()


def test_c3_emit_properties(c3):
    db, res = c3["db"], c3["res"]
    goff = c3["hs"].group_off.astype(np.int64)
    keep, emit = res.keep, res.emit.astype(np.int64)
    assert res.n_emit == int((keep != 0).sum()) == emit.size
    # pool by pool, pass 1 before pass 2 inside a pool, input order inside a pass: the sort key
    # (pool, pass, record) must be strictly increasing along the output -- which also makes every
    # record appear at most once
    gid = np.searchsorted(goff, emit, side="right") - 1
    key = (gid * 4 + keep[emit].astype(np.int64)) * (1 << 31) + (emit - goff[gid])
    assert (np.diff(key) > 0).all()
    assert (keep[emit] != 0).all()
    # all winners of a (pool, mate) share one score
    as_ = c3["hs"].as_
    pk = gid * 4 + keep[emit].astype(np.int64)
    starts = np.flatnonzero(np.r_[True, np.diff(pk) != 0])
    av = as_[emit]
    assert np.array_equal(np.maximum.reduceat(av, starts), np.minimum.reduceat(av, starts))


def test_c3_profile_properties(c3):
    st, ui, ab = c3["st"], c3["ui"].astype(np.int64), c3["ab"]
    assert st.insert_count == st.uniq_mapper_count + st.multi_mapper_count
    assert ui.sum() == 2 * st.uniq_mapper_count
    total = st.uniq_mapper_count + st.multi_mapper_count - st.purged_insert_count
    assert abs(ab.sum() - total) <= 1e-9 * total          # every non-purged multi-mapper adds exactly 1
    assert (ab >= ui / 2 - 1e-9).all() and 1 <= st.iterations <= 19
    # idempotence: a second finalize of the same counts
    ab2, st2 = c3["prof"].finalize()
    assert st2.iterations == st.iterations and np.allclose(ab, ab2, rtol=1e-9, atol=0)


def test_c3_prefix_separate_calls_with_1m_refs(c3):
    """200 000 pools, n_refs = 1 M, through msx_filter_enqueue + msx_profile_accumulate (not the fused
    call): equal to the oracle, and filter's part equal to the prefix of the big fused run."""
    import msamtools_amd as m
    ctx = c3["ctx"]
    db = m.DeviceBatch.synth(ctx, SEED, C3_PREFIX, C3_REFS, 4)
    run = m.FilterRun(ctx, db, **OPTS)
    prof = m.Profile(ctx, C3_REFS, "proportional")
    try:
        run.enqueue()
        run.finish()
        res = run.result()
        prof.accumulate(db, run.keep)
        ui = prof.ui()
        ab, st = prof.finalize()
        hs = m.HostSynth(SEED, C3_PREFIX, C3_REFS, 4)
        want = orc.run_filter(hs, **OPTS)
        assert np.array_equal(res.emit, want["emit"])
        ne = len(want["emit"])
        assert np.array_equal(c3["res"].emit[:ne], want["emit"])          # pools are independent
        ref = orc.run_profile(hs, C3_REFS, multi="proportional", sel=want["emit"])
        s = ref["stats"]
        assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
            (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
        assert np.array_equal(ui, ref["ui"])
        assert (st.iterations, st.converged) == (s.iterations, s.converged)
        want_ab = ref["abundance"]
        assert np.array_equal(ab == 0, want_ab == 0)
        assert (np.abs(ab - want_ab) / np.maximum(np.abs(want_ab), 1e-300)).max() <= REL_TOL
    finally:
        prof.close()
        run.free()
        db.free()


# ---- configs[3]: coverage ------------------------------------------------------------------------

def test_c4_coverage_full_size():
    import msamtools_amd as m
    ctx = m.Context(0)
    db = m.DeviceBatch.synth(ctx, SEED, C4_GROUPS, C4_REFS, 4)
    try:
        assert 47_000_000 < db.n_records < 53_000_000
        off = np.arange(C4_REFS + 1, dtype=np.int64) * C4_TLEN
        total = int(off[-1])
        d_off = ctx.alloc(off.nbytes)
        d_cov = ctx.alloc(4 * total + 8)
        d_seen = ctx.alloc(C4_REFS)
        ctx.to_dev(d_off, off)
        ctx.zero(d_cov, 4 * total + 8)
        ctx.zero(d_seen, C4_REFS)
        ctx.check(ctx.lib.msx_coverage_accumulate(ctx.h, C.byref(db.b), C.c_void_p(d_off), C4_REFS, total,
                                                  C.c_void_p(d_cov), C.c_void_p(d_seen)))
        ctx.check(ctx.lib.msx_coverage_finish(ctx.h, C.c_void_p(d_cov), total))
        cov = ctx.to_host(d_cov, total, np.int32)
        seen = ctx.to_host(d_seen, C4_REFS, np.uint8)
        # the whole-sample form (msx_coverage_depths: no zeroing, depths written once) on the same batch
        ctx.to_dev(d_cov, np.full(total + 2, 0x5a5a5a5a, np.uint32))
        ctx.zero(d_seen, C4_REFS)
        ctx.check(ctx.lib.msx_coverage_depths(ctx.h, C.byref(db.b), C.c_void_p(d_off), C4_REFS, total, C.c_void_p(d_cov),
                                              C.c_void_p(d_seen)))
        cov2 = ctx.to_host(d_cov, total, np.int32)
        seen2 = ctx.to_host(d_seen, C4_REFS, np.uint8)
        assert np.array_equal(cov, cov2) and np.array_equal(seen, seen2)
        del cov2
        # the command line's form: the sample batch after batch (msx_coverage_collect: the batches' pieces stay on the device,
        # msx_coverage_collect_finish sorts and sums them once) -- the same stream cut into seven uneven batches
        ctx.to_dev(d_cov, np.full(total + 2, 0x5a5a5a5a, np.uint32))
        ctx.zero(d_seen, C4_REFS)
        cuts = [0, 1_000_000, 1_000_017, 3_500_000, 3_500_000 + 9_000, 6_000_000, 9_999_999, C4_GROUPS]
        n_rec = 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = m.DeviceBatch.synth(ctx, SEED, hi - lo, C4_REFS, 4, first_group=lo)
            n_rec += part.n_records
            ctx.check(ctx.lib.msx_coverage_collect(ctx.h, C.byref(part.b), C.c_void_p(d_off), C4_REFS, total, C.c_void_p(d_cov),
                                                   C.c_void_p(d_seen)))
            part.free()
        assert n_rec == db.n_records
        n_streamed = C.c_int64(-1)
        ctx.check(ctx.lib.msx_coverage_collect_finish(ctx.h, C.c_void_p(d_cov), total, C.byref(n_streamed)))
        ctx.sync()
        assert n_streamed.value == 0
        cov3 = ctx.to_host(d_cov, total, np.int32)
        seen3 = ctx.to_host(d_seen, C4_REFS, np.uint8)
        assert np.array_equal(cov, cov3) and np.array_equal(seen, seen3)
        del cov3
        ctx.free(d_off), ctx.free(d_cov), ctx.free(d_seen)

        hs = HostCopy(db)
        # depth_sum == sum of the M/=/X run lengths of the mapped records (msam_coverage.c:63-74)
        op, w = hs.cigar & 0xF, (hs.cigar >> 4).astype(np.int64)
        n_cig = np.diff(hs.cigar_off.astype(np.int64))
        mapped = np.repeat(hs.tid >= 0, n_cig)
        want_sum = int(w[mapped & ((op == 0) | (op == 7) | (op == 8))].sum())
        assert int(cov.astype(np.int64).sum()) == want_sum
        assert (cov >= 0).all()
        # every per-base depth against the oracle's pile-up (adds 1 per base)
        ref = np.concatenate(orc.coverage(hs, [C4_TLEN] * C4_REFS))
        assert np.array_equal(cov, ref)
        assert np.array_equal(seen != 0, np.bincount(hs.tid[hs.tid >= 0], minlength=C4_REFS) > 0)
    finally:
        db.free()
        ctx.close()
