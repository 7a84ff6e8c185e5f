"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the reference's golden expectations.  Bar: bit-exact for record selection,
output order, integer statistics and counts; <= 1e-6 relative (stated in each
test) for the double-precision proportional profile (BASELINE.json north_star).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
import samio
from conftest import GOLDEN, fixture_path

pytestmark = pytest.mark.gpu

EXP = json.load(open(os.path.join(GOLDEN, "reference_expectations.json")))
REL_TOL = 1e-6      # north_star tolerance for the floating-point profile


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    c = m.Context(0)
    yield c
    c.close()


def records_str(rec, emit):
    return ",".join(f"{rec.name(i)}:{int(rec.flag[i])}" for i in emit)


def gpu_filter(ctx, rec, **opts):
    import msamtools_amd as m
    best = opts.get("besthit") or opts.get("uniqhit")
    goff = m.filter_pools(rec) if best else None
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    try:
        return m.run_filter(ctx, batch, **opts)
    finally:
        batch.free()


def gpu_profile(ctx, rec, n_features, multi, goff, keep=None, fmap=None):
    import msamtools_amd as m
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    prof = m.Profile(ctx, n_features, multi, fmap)
    kp = None
    try:
        if keep is not None:
            kp = ctx.alloc(max(len(keep), 1))
            ctx.to_dev(kp, np.ascontiguousarray(keep, dtype=np.uint8))
        prof.accumulate(batch, kp)
        ui = prof.ui()
        ab, st = prof.finalize()
        return ab.copy(), st, ui
    finally:
        if kp:
            ctx.free(kp)
        prof.close()
        batch.free()


def assert_profile_close(ab, st, ref, tol=REL_TOL):
    s = ref["stats"]
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
    want = ref["abundance"]
    assert ((ab == 0) == (want == 0)).all()
    scale = np.maximum(np.abs(want), 1e-300)
    assert (np.abs(ab - want) / scale).max() <= tol


# ---- the reference's golden vectors through the GPU ---------------------------

@pytest.mark.parametrize("case", EXP["filter"], ids=[c["name"] for c in EXP["filter"]])
def test_filter_golden_gpu(ctx, case):
    hdr, rec = samio.read_sam(fixture_path(case["fixture"]))
    res = gpu_filter(ctx, rec, **case["opts"])
    assert records_str(rec, res.emit) == case["records"], case["src"]
    want = orc.run_filter(rec, **case["opts"])
    assert res.emit.tolist() == want["emit"].tolist()
    assert res.n_emit == int((res.keep != 0).sum())
    if case["opts"].get("rescore"):
        mapped = (rec.flag & 4) == 0
        assert (res.as_out[mapped] == want["as_out"][mapped]).all()
        for key, val in case.get("as", {}).items():
            idx = [i for i in res.emit if f"{rec.name(i)}:{int(rec.flag[i])}" == key]
            assert idx and all(res.as_out[i] == val for i in idx)


def test_long_qname_besthit_gpu(ctx):
    blk = EXP["long_qname"]
    hdr, rec = samio.read_sam(fixture_path(blk["fixture"]))
    for case in blk["cases"]:
        res = gpu_filter(ctx, rec, **case["opts"])
        got = [[len(rec.name(i)), int(rec.flag[i]), hdr.target_name[rec.tid[i]], int(rec.as_[i])] for i in res.emit]
        assert got == blk["expected"]


@pytest.mark.parametrize("case", EXP["profile"], ids=[c["name"] for c in EXP["profile"]])
def test_profile_golden_gpu(ctx, case):
    import msamtools_amd as m
    hdr, rec = samio.read_sam(fixture_path(case["fixture"]))
    keep, sel = None, None
    if "filter_opts" in case:
        # the pipe `filter ... | profile -`: profile sees filter's output stream
        res = gpu_filter(ctx, rec, **dict(case["filter_opts"]))
        keep, sel = res.keep, res.emit
        goff = m.filter_pools(rec)
    else:
        goff = m.profile_pools(rec)
    ab, st, ui = gpu_profile(ctx, rec, hdr.n_targets, case["multi"], goff, keep)
    ref = orc.run_profile(rec, hdr.n_targets, multi=case["multi"], sel=sel)
    assert (ui == ref["ui"]).all()
    assert_profile_close(ab, st, ref)
    for key, fld in (("mapped", "insert_count"), ("multi_mapped", "multi_mapper_count"),
                     ("uniq_mapped", "uniq_mapper_count")):
        if key in case:
            assert getattr(st, fld) == case[key], (key, case["src"])
    ost = orc.OrcProfileStats()
    for f in ("insert_count", "uniq_mapper_count", "multi_mapper_count", "purged_insert_count"):
        setattr(ost, f, getattr(st, f))
    vals, purged, eff = orc.profile_finish(ab, hdr.target_len, ost, unit=case["unit"], nolen=case["nolen"],
                                           total=case["total"], mincount=case.get("mincount", -1),
                                           multi=case["multi"])
    names = ["Unknown"] + hdr.target_name
    for feat, (want, tol) in case["values"].items():
        assert abs(vals[names.index(feat)] - want) <= tol, (feat, case["src"])


def test_tiny_aln_gpu(ctx):
    """BASELINE.json configs[0]: tiny_aln.bam through filter -l 80 -p 95 -z 80 --besthit | profile."""
    import msamtools_amd as m
    t = EXP["tiny_aln"]
    hdr, rec = samio.read_bam(fixture_path(t["fixture"]))
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    st = m.aln_stats(ctx, batch)
    for k in ("length", "qlen", "qclip", "edit"):
        assert st[k].tolist() == t[k]
    res = m.run_filter(ctx, batch, **t["filter_opts"])
    assert res.emit.tolist() == t["emit"]
    uq = dict(t["filter_opts"], besthit=False, uniqhit=True)
    assert m.run_filter(ctx, batch, **uq).emit.tolist() == t["uniqhit_emit"]
    batch.free()
    ab, ps, ui = gpu_profile(ctx, rec, hdr.n_targets, "proportional", goff, res.keep)
    assert (ps.insert_count, ps.uniq_mapper_count, ps.multi_mapper_count, ps.purged_insert_count) == \
        (t["mapped"], t["uniq_mapped"], t["multi_mapped"], t["purged"])
    assert ps.iterations == 1 and ps.converged == 1 and ps.delta[1] == 0.0
    ref = orc.run_profile(rec, hdr.n_targets, sel=np.array(t["emit"], np.int32))
    assert (ab == ref["abundance"]).all()       # nothing is shared here: exact


def test_coverage_golden_gpu(ctx):
    import msamtools_amd as m
    blk = EXP["coverage"]
    hdr, rec = samio.read_sam(fixture_path(blk["fixture"]))
    batch = m.DeviceBatch.upload(ctx, rec, None)
    cov = m.coverage(ctx, batch, hdr.target_len)
    batch.free()
    for t, name in enumerate(hdr.target_name):
        assert cov[t].tolist() == blk["positions"][name]


# ---- error behaviour ---------------------------------------------------------------

def _mk(cigars, mds, nms=None, flags=None, names=None, as_=None, tids=None):
    b = samio._Builder()
    for i in range(len(cigars)):
        b.add((names[i] if names else f"r{i}").encode(), flags[i] if flags else 0,
              tids[i] if tids else 0, 0, samio.parse_cigar_text(cigars[i]),
              None if mds[i] is None else mds[i].encode(),
              None if nms is None else nms[i], None if as_ is None else as_[i])
    return b.build()


def test_fatal_errors_match_reference_messages(ctx):
    import msamtools_amd as m
    rec = _mk(["10M", "10M"], [None, None], nms=[0, None], names=["a", "b"])
    with pytest.raises(m.MsxError) as ei:
        gpu_filter(ctx, rec, l=5)
    assert ei.value.code == 1 and "Either NM or MD must be present" in ei.value.text
    with pytest.raises(m.MsxError) as ei:
        gpu_filter(ctx, rec, besthit=True)
    assert ei.value.code == 2 and "Required field AS not found" in ei.value.text
    batch = m.DeviceBatch.upload(ctx, rec, None)
    with pytest.raises(m.MsxError) as ei:
        m.run_filter(ctx, batch)
    assert ei.value.code == 3 and "requires atleast one of" in ei.value.text
    batch.free()


def test_unpinned_semantics_match_oracle(ctx):
    """'^' deletions in MD, odd CIGAR ops, mate-bit corner cases, unmapped flush rule."""
    import msamtools_amd as m
    rec = _mk(["50M2D48M", "10M", "10M", "5M1D5M"], ["50^AC48", "A9", "0A9", "5^A0T4"])
    batch = m.DeviceBatch.upload(ctx, rec, None)
    st = m.aln_stats(ctx, batch)
    batch.free()
    assert st["edit"].tolist() == [2, 0, 1, 2] and st["length"].tolist() == [100, 10, 10, 11]
    cig = [(5 << 4) | 0, (3 << 4) | 9, (2 << 4) | 2, (4 << 4) | 3, (1 << 4) | 6]
    b = samio._Builder()
    b.add(b"a", 0, 0, 0, cig, None, 1, None)
    b.add(b"b", 0, 0, 0, cig, b"5", None, None)
    rec = b.build()
    batch = m.DeviceBatch.upload(ctx, rec, None)
    st = m.aln_stats(ctx, batch)
    batch.free()
    assert st["length"].tolist() == [10, 7] and st["qlen"].tolist() == [5, 5] and st["edit"].tolist() == [1, 2]
    rec = _mk(["10M"] * 4, ["10"] * 4, flags=[0, 65, 129, 193], names=["q"] * 4, as_=[50, 10, 20, 99])
    assert gpu_filter(ctx, rec, besthit=True).emit.tolist() == [1, 2]
    rec = _mk(["10M", "*", "10M", "10M"], ["10", None, "10", "10"], flags=[0, 4, 256, 0],
              names=["a", "b", "a", "c"], as_=[10, 0, 20, 5])
    assert gpu_filter(ctx, rec, besthit=True).emit.tolist() == [0, 2, 3]


def test_empty_and_single_record_batches(ctx):
    import msamtools_amd as m
    hdr, rec = samio.read_sam(fixture_path("profile_empty.sam"))
    assert rec.n == 0
    res = gpu_filter(ctx, rec, l=10, besthit=True)
    assert res.n_emit == 0 and res.emit.size == 0
    rec = _mk(["100M"], ["100"], as_=[100], names=["solo"])
    res = gpu_filter(ctx, rec, p=95, besthit=True)
    assert res.emit.tolist() == [0]
    ab, st, ui = gpu_profile(ctx, rec, 3, "proportional", m.profile_pools(rec))
    assert ab.tolist() == [1.0, 0.0, 0.0] and st.insert_count == 1 and st.uniq_mapper_count == 1


# ---- seeded synthetic parity at oracle-friendly sizes ----------------------------------

SYNTH_CASES = [
    dict(l=80, p=95, z=80, besthit=True),
    dict(l=80, p=95, z=80, uniqhit=True),
    dict(p=97),
    dict(ppt=-980),
    dict(z=90, l=90),
    dict(p=99, invert=True, keep_unmapped=True),
    dict(besthit=True),
    dict(rescore=True, besthit=True),
    dict(l=70, rescore=True, uniqhit=True),
]


@pytest.fixture(scope="module")
def synth(ctx):
    import msamtools_amd as m
    hs = m.HostSynth(13579, 60000, 2000, 4)
    db = m.DeviceBatch.synth(ctx, 13579, 60000, 2000, 4)
    yield hs, db
    db.free()


def test_device_synth_equals_host_twin(ctx, synth):
    hs, db = synth
    assert db.n_records == hs.n_records and db.n_groups == hs.n_groups
    d = db.to_host()
    for k in ("flag", "rflags", "tid", "pos", "cigar_off", "cigar", "md_off", "md", "nm", "as_", "group_off"):
        assert (d[k] == getattr(hs, k)).all(), k


def test_synth_stats_parity(ctx, synth):
    import msamtools_amd as m
    hs, db = synth
    got = m.aln_stats(ctx, db)
    want = orc.aln_stats(hs)
    for k in ("length", "qlen", "qclip", "edit", "status"):
        assert (got[k] == want[k]).all(), k


@pytest.mark.parametrize("opts", SYNTH_CASES, ids=[json.dumps(o, sort_keys=True) for o in SYNTH_CASES])
def test_synth_filter_parity(ctx, synth, opts):
    import msamtools_amd as m
    hs, db = synth
    res = m.run_filter(ctx, db, **opts)
    want = orc.run_filter(hs, **opts)
    assert want["rc"] == 0
    assert res.n_emit == len(want["emit"])
    assert (res.emit == want["emit"]).all()
    kept = np.zeros(hs.n_records, bool)
    kept[want["emit"]] = True
    assert ((res.keep != 0) == kept).all()
    if opts.get("rescore"):
        assert (res.as_out == want["as_out"]).all()


@pytest.mark.parametrize("multi", ["proportional", "equal", "all", "ignore"])
def test_synth_fused_filter_profile_parity(ctx, synth, multi):
    """filter -l 80 -p 95 -z 80 --besthit | profile --multi=<mode> on the device vs the oracle pipe."""
    import msamtools_amd as m
    hs, db = synth
    opts = dict(l=80, p=95, z=80, besthit=True)
    run = m.FilterRun(ctx, db, **opts)
    run.enqueue()
    run.finish()
    prof = m.Profile(ctx, 2000, multi)
    prof.accumulate(db, run.keep)
    ui = prof.ui()
    ab, st = prof.finalize()
    ab = ab.copy()
    prof.close()
    run.free()
    sel = orc.run_filter(hs, **opts)["emit"]
    ref = orc.run_profile(hs, 2000, multi=multi, sel=sel)
    assert (ui == ref["ui"]).all()                     # integer counts: exact
    assert_profile_close(ab, st, ref)                  # doubles: <= 1e-6 relative
    if multi == "proportional":
        assert st.iterations == ref["stats"].iterations and st.converged == ref["stats"].converged
        assert abs(st.delta[st.iterations] - ref["stats"].last_delta) <= 1e-6 * max(ref["stats"].last_delta, 1e-30)


@pytest.mark.parametrize("multi", ["proportional", "equal", "all", "ignore"])
@pytest.mark.parametrize("opts", [dict(l=80, p=95, z=80, besthit=True), dict(uniqhit=True), dict(l=60, p=90)],
                         ids=["lpz_besthit", "uniqhit", "plain"])
def test_one_call_filter_profile_equals_two_calls_and_oracle(ctx, synth, multi, opts):
    """msx_filter_profile_enqueue == msx_filter_enqueue + msx_profile_accumulate (and the oracle pipe)."""
    import msamtools_amd as m
    hs, db = synth
    run = m.FilterRun(ctx, db, **opts)
    prof = m.Profile(ctx, 2000, multi)
    run.enqueue_with_profile(prof)
    run.finish()
    res = run.result()
    ui = prof.ui()
    ab, st = prof.finalize()
    ab = ab.copy()
    prof.close()
    # two calls
    run2 = m.FilterRun(ctx, db, **opts)
    run2.enqueue()
    run2.finish()
    res2 = run2.result()
    prof2 = m.Profile(ctx, 2000, multi)
    prof2.accumulate(db, run2.keep)
    ui2 = prof2.ui()
    ab2, st2 = prof2.finalize()
    assert (res.keep == res2.keep).all() and (res.emit == res2.emit).all() and res.n_emit == res2.n_emit
    assert (ui == ui2).all()
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
        (st2.insert_count, st2.uniq_mapper_count, st2.multi_mapper_count, st2.purged_insert_count)
    assert (np.abs(ab - ab2) <= 1e-9 * np.maximum(np.abs(ab2), 1e-300)).all()
    prof2.close()
    run.free()
    run2.free()
    ref_f = orc.run_filter(hs, **opts)
    assert (res.emit == ref_f["emit"]).all()
    ref = orc.run_profile(hs, 2000, multi=multi, sel=ref_f["emit"])
    assert (ui == ref["ui"]).all()
    assert_profile_close(ab, st, ref)


def test_synth_profile_subcommand_with_fmap(ctx, synth):
    """`profile` on the raw stream (keep == NULL) with a tid -> feature map (--genome)."""
    import msamtools_amd as m
    hs, db = synth
    fmap = (np.arange(2000, dtype=np.int32) // 7).astype(np.int32)
    nf = int(fmap.max()) + 1
    prof = m.Profile(ctx, nf, "proportional", fmap)
    prof.accumulate(db, None)
    ui = prof.ui()
    ab, st = prof.finalize()
    ab = ab.copy()
    prof.close()
    ref = orc.run_profile(hs, nf, multi="proportional", fmap=fmap)
    assert (ui == ref["ui"]).all()
    assert_profile_close(ab, st, ref)


def test_split_proportional_api_matches_finalize(ctx, synth):
    """begin / (local, apply)* / purged -- the form a multi-GPU caller drives -- equals finalize."""
    import msamtools_amd as m
    hs, db = synth
    prof = m.Profile(ctx, 2000, "proportional")
    prof.accumulate(db, None)
    ab1, st1 = prof.finalize()
    ab1 = ab1.copy()
    prof.prop_begin()
    k, delta = 0, 1.0
    while k < 19:
        prof.prop_local()
        delta = prof.prop_apply()
        k += 1
        if delta < 1e-10:
            break
    purged = prof.prop_purged()
    ab2 = ctx.to_host(prof.abundance_ptr(), 2000, np.float64)
    prof.close()
    assert k == st1.iterations and purged == st1.purged_insert_count
    assert np.allclose(ab1, ab2, rtol=1e-9, atol=0)


def test_streaming_batches_equal_one_batch(ctx):
    """Accumulating the stream in several QNAME-aligned batches gives the same profile."""
    import msamtools_amd as m
    whole = m.DeviceBatch.synth(ctx, 24680, 30000, 500, 4)
    p1 = m.Profile(ctx, 500, "proportional")
    p1.accumulate(whole, None)
    ui1 = p1.ui()
    ab1, st1 = p1.finalize()
    ab1 = ab1.copy()
    p2 = m.Profile(ctx, 500, "proportional")
    for first in (0, 10000, 20000):
        part = m.DeviceBatch.synth(ctx, 24680, 10000, 500, 4, first_group=first)
        p2.accumulate(part, None)
        ctx.sync()
        part.free()
    ui2 = p2.ui()
    ab2, st2 = p2.finalize()
    assert (ui1 == ui2).all()
    assert (st1.insert_count, st1.uniq_mapper_count, st1.multi_mapper_count, st1.purged_insert_count) == \
        (st2.insert_count, st2.uniq_mapper_count, st2.multi_mapper_count, st2.purged_insert_count)
    assert np.allclose(ab1, ab2, rtol=1e-9, atol=0)
    p1.close()
    p2.close()
    whole.free()


def test_streaming_one_call_filter_profile_equals_one_batch(ctx):
    """msx_filter_profile_enqueue batch by batch (the command-line pipe's shape) == one batch."""
    import msamtools_amd as m
    opts = dict(l=80, p=95, z=80, besthit=True)
    whole = m.DeviceBatch.synth(ctx, 97531, 30000, 500, 4)
    r1 = m.FilterRun(ctx, whole, **opts)
    p1 = m.Profile(ctx, 500, "proportional")
    r1.enqueue_with_profile(p1)
    r1.finish()
    ui1 = p1.ui()
    ab1, st1 = p1.finalize()
    ab1 = ab1.copy()
    p2 = m.Profile(ctx, 500, "proportional")
    n_emit = 0
    for first in (0, 7000, 19000):
        part = m.DeviceBatch.synth(ctx, 97531, (7000, 12000, 11000)[(0, 7000, 19000).index(first)], 500, 4, first_group=first)
        r2 = m.FilterRun(ctx, part, **opts)
        r2.enqueue_with_profile(p2)
        n_emit += int(r2.finish().n_emit)
        r2.free()
        part.free()
    assert n_emit == int(r1.status.n_emit)
    assert (ui1 == p2.ui()).all()
    ab2, st2 = p2.finalize()
    assert (st1.insert_count, st1.uniq_mapper_count, st1.multi_mapper_count, st1.purged_insert_count) == \
        (st2.insert_count, st2.uniq_mapper_count, st2.multi_mapper_count, st2.purged_insert_count)
    assert np.allclose(ab1, ab2, rtol=1e-9, atol=0)
    for x in (p1, p2):
        x.close()
    r1.free()
    whole.free()


@pytest.mark.parametrize("multi", ["proportional", "equal"])
def test_finalize_straight_after_the_fused_enqueue(ctx, multi):
    """msx_filter_profile_enqueue leaves its side lanes (unique-insert counts, filter's output order) running;
    msx_profile_finalize builds the sharing store beside them and joins before it needs the counts, and
    msx_filter_finish -- called only afterwards here, as bench.py does -- still sees the whole output order.
    Same numbers as the step-by-step sequence."""
    import msamtools_amd as m
    opts = dict(l=80, p=95, z=80, besthit=True)
    batch = m.DeviceBatch.synth(ctx, 24680, 60000, 2000, 4)
    ref_run = m.FilterRun(ctx, batch, **opts)
    ref_run.enqueue()
    ref_run.finish()
    want = ref_run.result()
    p1 = m.Profile(ctx, 2000, multi)
    p1.accumulate(batch, ref_run.keep)
    ui1 = p1.ui()
    ab1, st1 = p1.finalize()
    ab1 = ab1.copy()
    for _ in range(3):                    # repeated: a later call must not trip over lanes an earlier one left behind
        p2 = m.Profile(ctx, 2000, multi)
        run = m.FilterRun(ctx, batch, **opts)
        run.enqueue_with_profile(p2)
        ab2, st2 = p2.finalize()          # no msx_filter_finish in between
        run.finish()
        got = run.result()
        assert got.n_emit == want.n_emit
        assert (got.keep == want.keep).all() and (got.emit == want.emit).all()
        assert (p2.ui() == ui1).all()
        assert (st1.insert_count, st1.uniq_mapper_count, st1.multi_mapper_count, st1.purged_insert_count, st1.iterations) == \
            (st2.insert_count, st2.uniq_mapper_count, st2.multi_mapper_count, st2.purged_insert_count, st2.iterations)
        assert np.allclose(ab1, ab2, rtol=1e-9, atol=0)
        run.free()
        p2.close()
    p1.close()
    ref_run.free()
    batch.free()


def test_ragged_inputs(ctx):
    """Very long CIGAR/MD payloads (beyond the LDS staging tile), a 5000-record pool,
    pools with > 4 distinct references."""
    import msamtools_amd as m
    rng = np.random.default_rng(7)
    b = samio._Builder()
    # 300 records with 40-op CIGARs and long MD strings -> tile payload > LDS capacity
    for i in range(300):
        ops = []
        for j in range(20):
            ops += [(int(rng.integers(1, 9)) << 4) | 0, (1 << 4) | (1 if j % 2 else 2)]
        md = "".join(f"{int(rng.integers(0, 30))}{'ACGT'[int(rng.integers(0, 4))]}" for _ in range(30)) + "5"
        b.add(f"long{i // 3}".encode(), 0 if i % 3 == 0 else 256, int(rng.integers(0, 50)), 0, ops, md.encode(), None,
              int(rng.integers(0, 200)))
    # one pool of 5000 records over 37 references with many score ties
    for i in range(5000):
        b.add(b"huge", (65 if i % 2 else 129) | (256 if i > 1 else 0), int(rng.integers(0, 37)), 0,
              [(100 << 4) | 0], b"100", None, int(rng.integers(90, 100)))
    # pools with 6..12 distinct references
    for g in range(200):
        k = int(rng.integers(6, 13))
        for j in range(k):
            b.add(f"multi{g}".encode(), 0 if j == 0 else 256, int(rng.integers(0, 50)), 0, [(100 << 4) | 0], b"100",
                  None, 100)
    rec = b.build()
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    got = m.aln_stats(ctx, batch)
    want = orc.aln_stats(rec)
    for k in ("length", "qlen", "qclip", "edit"):
        assert (got[k] == want[k]).all(), k
    for opts in (dict(besthit=True), dict(uniqhit=True), dict(l=50, p=60, besthit=True)):
        res = m.run_filter(ctx, batch, **opts)
        w = orc.run_filter(rec, **opts)
        assert (res.emit == w["emit"]).all() and res.n_emit == len(w["emit"])
    batch.free()
    for multi in ("proportional", "equal", "all"):
        ab, st, ui = gpu_profile(ctx, rec, 50, multi, m.profile_pools(rec))
        ref = orc.run_profile(rec, 50, multi=multi)
        assert (ui == ref["ui"]).all()
        assert_profile_close(ab, st, ref)


def test_extreme_scores_and_mate_bits(ctx):
    """Corners no reference fixture pins (source-derived, msam_filter.c:192-263): AS = INT32_MIN alone in a pool (the
    reference writes it: best_count goes 0 -> 1 through the score == best_score branch, :225-229), tied, and beside
    larger scores; AS of BAM type I above 2^31 (bam_aux2i's int64 truncated to int32_t: a negative score); paired pools
    whose records carry both mate bits, neither, or a mix -- kernel against the oracle, --besthit and --uniqhit, and
    the inserts profile counts from what is written."""
    import msamtools_amd as m
    lo = -(1 << 31)
    b = samio._Builder()
    cig = [(100 << 4) | 0]

    def pool(name, scores, flags=None, tids=None):
        for i, sc in enumerate(scores):
            fl = flags[i] if flags else (0 if i == 0 else 256)
            b.add(name, fl, tids[i] if tids else i % 7, 10 * i, cig, b"100", None, sc)
    pool(b"min_alone", [lo])
    pool(b"min_tied", [lo, lo, lo])
    pool(b"min_and_more", [lo, lo + 1, 5, lo])
    pool(b"min_vs_zero", [0, lo])
    pool(b"big_I", [0x80000005 - (1 << 32), 0xfffffff0 - (1 << 32), 0x7fffffff])       # type I values as int32
    pool(b"max_tied", [0x7fffffff, 0x7fffffff, lo])
    pool(b"both_bits", [50, 60, 60, 40], flags=[0xC1, 0xC1 | 256, 0xC1 | 256, 0xC1 | 256])
    pool(b"both_and_one", [50, 60, 70, 70, lo], flags=[0xC1, 0x41 | 256, 0x81 | 256, 0xC1 | 256, 0x81 | 256])
    pool(b"neither_bit_paired", [30, 30, 20], flags=[0x01, 0x01 | 256, 0x01 | 256])
    pool(b"unpaired_with_mate_bits", [30, 31, 31], flags=[0x40, 0x80 | 256, 0xC0 | 256])
    pool(b"min_paired", [lo, lo, lo, 3], flags=[0x41, 0x81 | 256, 0x41 | 256, 0x81 | 256], tids=[1, 1, 2, 3])
    rec = b.build()
    for opts in (dict(besthit=True), dict(uniqhit=True), dict(l=50, besthit=True), dict(p=90, uniqhit=True)):
        got = gpu_filter(ctx, rec, **opts)
        want = orc.run_filter(rec, **opts)
        assert want["rc"] == 0
        assert got.emit.tolist() == want["emit"].tolist(), opts
    # the inserts of `filter --besthit | profile`, fused call against the oracle run as the two commands
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff, filter_pools=True)
    run = m.FilterRun(ctx, batch, besthit=True)
    prof = m.Profile(ctx, 7, "proportional")
    run.enqueue_with_profile(prof)
    run.finish()
    sel = run.result().emit
    ab, st = prof.finalize()
    ref = orc.run_profile(rec, 7, multi="proportional", sel=sel)
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count) == \
        (ref["stats"].insert_count, ref["stats"].uniq_mapper_count, ref["stats"].multi_mapper_count)
    assert_profile_close(ab, st, ref)
    prof.close()
    run.free()
    batch.free()


def test_int32_wraparound_matches_oracle(ctx):
    """Alignments > 2.1 Mbp overflow the reference's int32 threshold products; the kernel wraps like the oracle."""
    import msamtools_amd as m
    big = (1 << 28) - 1
    b = samio._Builder()
    b.add(b"a", 0, 0, 0, [(big << 4) | 0] * 9, None, 5, None)
    b.add(b"b", 0, 0, 0, [(3000000 << 4) | 0, (100 << 4) | 4], None, 30000, None)
    rec = b.build()
    for opts in (dict(p=95), dict(z=50), dict(l=1000), dict(ppt=-900)):
        assert gpu_filter(ctx, rec, **opts).emit.tolist() == orc.run_filter(rec, **opts)["emit"].tolist()


def test_synth_coverage_parity(ctx, synth):
    """Per-base depth (difference array + prefix sum) equals the oracle's per-base counting."""
    import msamtools_amd as m
    hs, db = synth
    tlen = [5000] * 2000
    cov, touched, dsum = m.coverage(ctx, db, tlen, summary=True)
    want = orc.coverage(hs, tlen)
    for t in range(len(tlen)):
        assert (cov[t] == want[t]).all(), t
    assert sum(int(c.sum()) for c in cov) > 0
    # msx_coverage_summary: the two sums mWriteCoverageSummaryToStream divides by the target's length (msam_coverage.c:188-219)
    assert touched.tolist() == [int((w != 0).sum()) for w in want]
    assert dsum.tolist() == [int(w.astype(np.int64).sum()) for w in want]


@pytest.mark.parametrize("n_groups,n_refs,tl", [(40000, 2000, 5000), (300000, 300, 5000), (200000, 4, 5000), (30000, 50000, 5000)])
def test_whole_sample_coverage_parity(ctx, n_groups, n_refs, tl):
    """msx_coverage_depths (marks sorted by (sign, tile), every tile's depths finished where its marks are gathered; hot
    tiles pre-reduced; runs behind a D / N through the overflow lists) equals the oracle's per-base counting and the
    streamed path -- few references: every tile is a hot one; many short ones: tiles span dozens of targets"""
    import msamtools_amd as m
    hs = m.HostSynth(4242, n_groups, n_refs, 4)
    db = m.DeviceBatch.synth(ctx, 4242, n_groups, n_refs, 4)
    tlen = [tl] * n_refs
    want = orc.coverage(hs, tlen)
    # the form with one word per run piece (4 K-cell tiles, the tile's top byte beside the key), then the form with a +1
    # and a -1 mark per run (MSX_COV_MARKS=1: what samples beyond 255 * 2^20 cells take)
    for env in (None, "MSX_COV_MARKS"):
        if env:
            os.environ[env] = "1"
        try:
            got = m.coverage(ctx, db, tlen, whole_sample=True)
        finally:
            if env:
                del os.environ[env]
        for t in range(n_refs):
            assert (got[t] == want[t]).all(), (env, t)
    db.free()


def test_whole_sample_coverage_long_runs(ctx):
    """Runs longer than a tile (a contig aligned end to end: cut into one piece per 4 K-cell tile, all but the first through
    the overflow lists) and runs behind long N skips -- both whole-sample forms and the streamed path against the oracle's
    per-base counting.  (Every run stays inside its target: beyond it the reference writes outside its array,
    msam_coverage.c:66-70, and so does the oracle.)"""
    import msamtools_amd as m
    hs = m.HostSynth(777, 20000, 40, 4)
    assert hs.n_records >= 1 << 16                     # (below that msx_coverage_depths takes the streamed path)
    cig = hs.cigar
    first = hs.cigar_off[:-1]
    nops = np.diff(hs.cigar_off)
    r = np.arange(hs.n_records)
    long_m = first[(r % 211 == 0)]
    cig[long_m] = (cig[long_m] & 15) | (9000 << 4)                      # 9000M: three tiles
    very = first[(r % 4099 == 0)]
    cig[very] = (cig[very] & 15) | (20000 << 4)                         # five or six tiles
    skip = first[(nops == 3) & (r % 7 == 0)] + 1
    cig[skip] = 3 | (6000 << 4)                                         # xM 6000N yM
    tlen = [30000] * 40
    want = orc.coverage(hs, tlen)
    db = m.DeviceBatch.upload(ctx, hs)
    for env in (None, "MSX_COV_MARKS", "MSX_COV_STREAMED"):
        if env:
            os.environ[env] = "1"
        try:
            got = m.coverage(ctx, db, tlen, whole_sample=True)
        finally:
            if env:
                del os.environ[env]
        for t in range(40):
            assert (got[t] == want[t]).all(), (env, t, np.flatnonzero(got[t] != want[t])[:5])
    db.free()


@pytest.mark.parametrize("seed,n_refs,tl", [(1, 7, 40000), (2, 300, 3000), (3, 20000, 700)])
def test_whole_sample_coverage_fuzz(ctx, seed, n_refs, tl):
    """Random CIGARs (every operation code, 1-9 operations, zero-length ones among them), random targets (unmapped records
    too) and positions such that every run stays inside its target: both whole-sample forms and the streamed path
    against the oracle's per-base counting (msam_coverage.c:33-87)."""
    import msamtools_amd as m
    rng = np.random.default_rng(seed)
    hs = m.HostSynth(99 + seed, 16000, max(n_refs, 4), 4)
    n = hs.n_records
    assert n >= 1 << 16
    nops = rng.integers(1, 10, n)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(nops)
    ops = rng.choice(np.arange(9), size=int(off[-1]), p=[0.45, 0.1, 0.12, 0.08, 0.08, 0.03, 0.02, 0.06, 0.06]).astype(np.uint32)
    wd = rng.integers(0, 60, int(off[-1])).astype(np.uint32)
    wd[rng.random(wd.size) < 0.02] = rng.integers(300, 1500, int((rng.random(wd.size) < 0.02).sum()) or 1)[0]
    ref_span = lambda: np.add.reduceat(np.where(np.isin(ops, [0, 2, 3, 7, 8]), wd, 0).astype(np.int64), off[:-1].astype(np.int64))
    for i in np.flatnonzero(ref_span() > tl - 1):      # a record longer than its target: all soft clips but one 10M
        ops[off[i]:off[i + 1]] = 4
        ops[off[i]], wd[off[i]] = 0, 10
    span = ref_span()
    tid = rng.integers(-1, n_refs, n).astype(np.int32)
    pos = (rng.random(n) * (tl - span)).astype(np.int32)
    assert ((pos + span) <= tl).all()
    hs.cigar_off, hs.cigar = off, (wd << 4 | ops).astype(np.uint32)
    hs.tid, hs.pos = tid, pos
    tlen = [tl] * n_refs
    want = orc.coverage(hs, tlen)
    db = m.DeviceBatch.upload(ctx, hs)
    for env in (None, "MSX_COV_MARKS", "MSX_COV_STREAMED"):
        if env:
            os.environ[env] = "1"
        try:
            got = m.coverage(ctx, db, tlen, whole_sample=True)
        finally:
            if env:
                del os.environ[env]
        for t in range(n_refs):
            assert (got[t] == want[t]).all(), (env, t, np.flatnonzero(got[t] != want[t])[:5])
    db.free()


def _cov_fuzz_records(hs, seed, n_refs, tl):
    """hs with random CIGARs (every operation code, 1-9 operations), random targets and positions, runs inside their targets"""
    rng = np.random.default_rng(seed)
    n = hs.n_records
    nops = rng.integers(1, 10, n)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(nops)
    ops = rng.choice(np.arange(9), size=int(off[-1]), p=[0.45, 0.1, 0.12, 0.08, 0.08, 0.03, 0.02, 0.06, 0.06]).astype(np.uint32)
    wd = rng.integers(0, 60, int(off[-1])).astype(np.uint32)
    ref_span = lambda: np.add.reduceat(np.where(np.isin(ops, [0, 2, 3, 7, 8]), wd, 0).astype(np.int64), off[:-1].astype(np.int64))
    for i in np.flatnonzero(ref_span() > tl - 1):
        ops[off[i]:off[i + 1]] = 4
        ops[off[i]], wd[off[i]] = 0, 10
    span = ref_span()
    hs.cigar_off, hs.cigar = off, (wd << 4 | ops).astype(np.uint32)
    hs.tid = rng.integers(-1, n_refs, n).astype(np.int32)
    hs.pos = (rng.random(n) * (tl - span)).astype(np.int32)
    return hs


@pytest.mark.parametrize("n_groups,n_refs,tl,cuts", [
    (60000, 2000, 5000, [0.0, 0.31, 0.31, 0.32, 0.77, 1.0]),        # an empty batch, a small one
    (300000, 6, 5000, [0.0, 0.05, 0.5, 0.52, 1.0]),                 # every tile a hot one
    (20000, 50000, 5000, [0.0, 0.001, 0.002, 0.6, 1.0])])           # tiny batches (far below a sort tile), many targets
def test_collected_coverage_parity(ctx, n_groups, n_refs, tl, cuts):
    """msx_coverage_collect / msx_coverage_collect_finish -- the command line's form of the whole-sample pile-up: the batches'
    pieces kept on the device, sorted and summed once -- against the oracle's per-base counting (msam_coverage.c:33-87),
    the covered flags (:45-49) included; no batch takes the streamed way."""
    import msamtools_amd as m
    hs = m.HostSynth(515, n_groups, n_refs, 4)
    tlen = [tl] * n_refs
    want = orc.coverage(hs, tlen)
    n = hs.n_records
    edges = [int(round(c * n)) for c in cuts]
    parts = [m.DeviceBatch.upload(ctx, m.RecordSlice(hs, lo, hi)) for lo, hi in zip(edges[:-1], edges[1:])]
    try:
        got, n_streamed, flags = m.coverage_collected(ctx, parts, tlen, covered=True)
    finally:
        for p in parts:
            p.free()
    assert n_streamed == 0
    for t in range(n_refs):
        assert (got[t] == want[t]).all(), (t, np.flatnonzero(got[t] != want[t])[:5])
    assert np.array_equal(flags != 0, np.bincount(hs.tid[hs.tid >= 0], minlength=n_refs) > 0)
    # a second sample on the same context starts from nothing
    again, n_streamed = m.coverage_collected(ctx, [], tlen)
    assert n_streamed == 0 and all(int(a.sum()) == 0 and (a == 0).all() for a in again)


def test_collected_coverage_with_batches_that_go_the_streamed_way(ctx):
    """A batch whose overflow lists run full (most records with several runs behind D / N operations) is piled up the streamed
    way inside msx_coverage_collect -- marks in the depth array, zeroed then -- and the finish adds the other batches'
    tile depths to its prefix sums: plain batch, such a batch, plain batch; and the whole sample the streamed way
    (MSX_COV_STREAMED=1: what a sample of more than 255 x 2^20 cells takes)."""
    import msamtools_amd as m
    n_refs, tl = 300, 5000                                  # (the synthetic positions stay below 5000: msx_synth.h)
    a = m.HostSynth(31, 30000, n_refs, 4)
    b = _cov_fuzz_records(m.HostSynth(32, 16000, n_refs, 4), 7, n_refs, tl)
    c = m.HostSynth(33, 9000, n_refs, 4)
    tlen = [tl] * n_refs
    want = [x + y + z for x, y, z in zip(orc.coverage(a, tlen), orc.coverage(b, tlen), orc.coverage(c, tlen))]
    for env, min_streamed in ((None, 1), ("MSX_COV_STREAMED", 3)):
        parts = [m.DeviceBatch.upload(ctx, m.RecordSlice(x, 0, x.n_records)) for x in (a, b, c)]
        if env:
            os.environ[env] = "1"
        try:
            got, n_streamed = m.coverage_collected(ctx, parts, tlen)
        finally:
            if env:
                del os.environ[env]
            for p in parts:
                p.free()
        assert n_streamed >= min_streamed, (env, n_streamed)
        if env is None:
            assert n_streamed < 3
        for t in range(n_refs):
            assert (got[t] == want[t]).all(), (env, t, np.flatnonzero(got[t] != want[t])[:5])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_collected_coverage_random_cuts(ctx, seed):
    """The same records under random partitions into batches (1 to 40 batches, empty ones among them, random CIGARs with runs
    behind D / N operations): msx_coverage_collect + _finish give the depths of the one call, whatever the cuts and whichever
    batches went the streamed way."""
    import msamtools_amd as m
    rng = np.random.default_rng(1000 + seed)
    n_refs, tl = 500, 5000
    hs = m.HostSynth(60 + seed, 24000, n_refs, 4)
    if seed % 2 == 0:
        hs = _cov_fuzz_records(hs, seed, n_refs, tl)
    tlen = [tl] * n_refs
    whole = m.DeviceBatch.upload(ctx, m.RecordSlice(hs, 0, hs.n_records))
    want = m.coverage(ctx, whole, tlen, whole_sample=True)
    whole.free()
    for trial in range(3):
        k = int(rng.integers(1, 41))
        edges = np.sort(np.concatenate([[0, hs.n_records], rng.integers(0, hs.n_records + 1, k - 1)])).tolist()
        parts = [m.DeviceBatch.upload(ctx, m.RecordSlice(hs, lo, hi)) for lo, hi in zip(edges[:-1], edges[1:])]
        try:
            got, n_streamed = m.coverage_collected(ctx, parts, tlen)
        finally:
            for p in parts:
                p.free()
        for t in range(n_refs):
            assert (got[t] == want[t]).all(), (seed, trial, k, n_streamed, t)
