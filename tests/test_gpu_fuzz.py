"""Randomised adversarial parity: the HIP path vs the oracle on batches built to
hit the corners -- unmapped records that flush pools, missing MD/NM/AS tags,
every CIGAR op code 0..15, MD strings with '^', letters at the start, empty and
non-ASCII bytes, mate-bit combinations, huge and tiny pools, tid == -1 inside
pools, duplicate references -- under every option family."""
import numpy as np
import pytest

import oracle_lib as orc
import samio

pytestmark = pytest.mark.gpu

MD_ALPHABET = list(b"0123456789^ACGTN") + [0x7f, 0x80, 0xff, ord("a")]


def random_batch(seed, n_groups=1500, n_refs=37, with_tags="mixed"):
    rng = np.random.default_rng(seed)
    b = samio._Builder()
    for g in range(n_groups):
        size = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 8, 17, 70]))
        name = f"q{g}".encode()
        paired = rng.random() < 0.6
        for k in range(size):
            flag = 0
            if paired:
                flag |= 1
                flag |= int(rng.choice([0x40, 0x80, 0x40, 0x80, 0xC0, 0x00]))
            if k and rng.random() < 0.7:
                flag |= 0x100
            if rng.random() < 0.1:
                flag |= 0x800
            unmapped = rng.random() < 0.06
            if unmapped:
                flag |= 4
            tid = -1 if (unmapped and rng.random() < 0.7) or rng.random() < 0.02 else int(rng.integers(0, n_refs))
            ncig = int(rng.choice([0, 1, 1, 1, 2, 3, 5, 12]))
            cig = []
            for _ in range(ncig):
                op = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15]))
                cig.append((int(rng.integers(0, 160)) << 4) | op)
            r = rng.random()
            if with_tags == "all":
                has_md, has_nm, has_as = True, True, True
            else:
                has_md, has_nm, has_as = r < 0.6, (r > 0.3), rng.random() < 0.97
                if not has_md and not has_nm:
                    has_nm = True                     # the fatal path is tested separately
            md = None
            if has_md:
                ln = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 40]))
                md = bytes(int(rng.choice(MD_ALPHABET)) for _ in range(ln))
                md = md.replace(b"\0", b"0")
            nm = int(rng.integers(0, 40)) if has_nm else None
            as_ = int(rng.integers(-5, 12)) if has_as else None
            b.add(name, flag, tid, int(rng.integers(0, 1000)), cig, md, nm, as_)
    return b.build()


OPTS = [
    dict(l=20), dict(p=90), dict(ppt=-930), dict(z=60), dict(l=10, p=80, z=50),
    dict(p=95, invert=True), dict(p=95, invert=True, keep_unmapped=True), dict(ppt=-900, invert=True, keep_unmapped=True),
    dict(l=30, rescore=True),
]
BEST = [dict(besthit=True), dict(uniqhit=True), dict(l=15, p=70, besthit=True), dict(z=40, uniqhit=True),
        dict(rescore=True, besthit=True), dict(l=25, rescore=True, uniqhit=True)]


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    c = m.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_stats_and_plain_filter(ctx, seed):
    import msamtools_amd as m
    rec = random_batch(1000 + seed)
    batch = m.DeviceBatch.upload(ctx, rec, None)
    got = m.aln_stats(ctx, batch)
    want = orc.aln_stats(rec)
    for k in ("length", "qlen", "qclip", "edit", "status"):
        assert (got[k] == want[k]).all(), k
    for opts in OPTS:
        res = m.run_filter(ctx, batch, **opts)
        w = orc.run_filter(rec, **opts)
        assert w["rc"] == 0
        assert (res.emit == w["emit"]).all() and res.n_emit == len(w["emit"]), opts
        if opts.get("rescore"):
            mapped = (rec.flag & 4) == 0
            assert (res.as_out[mapped] == w["as_out"][mapped]).all()
    batch.free()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_besthit_and_fused_profile(ctx, seed):
    import msamtools_amd as m
    rec = random_batch(2000 + seed)
    # AS must exist on participating records or the reference dies: give every record one
    rec.rflags[:] |= samio.HAS_AS
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    for opts in BEST:
        run = m.FilterRun(ctx, batch, **opts)
        run.enqueue()
        run.finish()
        res = run.result()
        w = orc.run_filter(rec, **opts)
        assert w["rc"] == 0
        assert (res.emit == w["emit"]).all() and res.n_emit == len(w["emit"]), opts
        for multi in ("proportional", "equal", "all", "ignore"):
            prof = m.Profile(ctx, 37, multi)
            prof.accumulate(batch, run.keep)
            ui = prof.ui()
            ab, st = prof.finalize()
            ref = orc.run_profile(rec, 37, multi=multi, sel=w["emit"])
            assert (ui == ref["ui"]).all(), (opts, multi)
            s = ref["stats"]
            assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
                (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count), (opts, multi)
            want = ref["abundance"]
            assert ((ab == 0) == (want == 0)).all()
            assert (np.abs(ab - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all(), (opts, multi)
            prof.close()
            # the same pipe as one call
            run1 = m.FilterRun(ctx, batch, **opts)
            prof1 = m.Profile(ctx, 37, multi)
            run1.enqueue_with_profile(prof1)
            run1.finish()
            assert (run1.result().emit == w["emit"]).all(), (opts, multi)
            assert (prof1.ui() == ref["ui"]).all(), (opts, multi)
            ab1, st1 = prof1.finalize()
            assert (st1.insert_count, st1.uniq_mapper_count, st1.multi_mapper_count, st1.purged_insert_count) == \
                (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count), (opts, multi)
            assert (np.abs(ab1 - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all(), (opts, multi)
            prof1.close()
            run1.free()
        run.free()
    batch.free()


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_profile_subcommand(ctx, seed):
    """`profile` on the raw stream: tid == -1 records are transparent, pools by QNAME."""
    import msamtools_amd as m
    rec = random_batch(3000 + seed)
    goff = m.profile_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    fmap = (np.arange(37, dtype=np.int32) * 7 % 11).astype(np.int32)
    for multi in ("proportional", "equal", "all", "ignore"):
        for fm, nf in ((None, 37), (fmap, 11)):
            prof = m.Profile(ctx, nf, multi, fm)
            prof.accumulate(batch, None)
            ui = prof.ui()
            ab, st = prof.finalize()
            ref = orc.run_profile(rec, nf, multi=multi, fmap=fm)
            assert (ui == ref["ui"]).all()
            s = ref["stats"]
            assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
                (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
            want = ref["abundance"]
            assert (np.abs(ab - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all()
            prof.close()
    batch.free()


def test_fuzz_missing_as_is_fatal_at_first_offender(ctx):
    import msamtools_amd as m
    rec = random_batch(4242, n_groups=300, with_tags="all")
    victim = 500
    rec.rflags[victim] &= ~np.uint8(samio.HAS_AS)
    w = orc.run_filter(rec, besthit=True)
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    if w["rc"] == 2:
        with pytest.raises(m.MsxError) as ei:
            m.run_filter(ctx, batch, besthit=True)
        assert ei.value.code == 2
    else:      # the record did not take part in any best-hit pass: no error on either side
        assert (m.run_filter(ctx, batch, besthit=True).emit == w["emit"]).all()
    batch.free()
