"""One context, a random SEQUENCE of operations of random shapes, every one checked against the oracle (-m gpu).

The other parity tests run one kind of call per test, mostly on a context that has seen few shapes; what carries over from
call to call -- reserved scratch that is grown, kept and reused (msx_reserve, cov_grow_keep, the sort tables), flags and
lists left behind by the call before -- is what this walks through: filter (plain, best hit, fused with the profile in all
four --multi modes), statistics, coverage in its three forms (streamed, whole sample, collected in random cuts), BGZF
deflate -> inflate round trips at random levels, in random order and sizes from a dozen to a million records.  Round 5
found such a bug by accident (a whole-sample coverage call that went the overflow way left words the next call's search
walked into); MSX_POISON=1 (scratch filled with 0xA5 whenever it is handed out) is on for the whole run."""
import os
import zlib

import numpy as np
import pytest

import oracle_lib as orc
import samio
from test_gpu_fuzz import random_batch
from test_gpu_parity import _cov_fuzz_records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    old = os.environ.get("MSX_POISON")
    os.environ["MSX_POISON"] = "1"
    c = m.Context(0)
    yield c
    c.close()
    if old is None:
        del os.environ["MSX_POISON"]
    else:
        os.environ["MSX_POISON"] = old


def op_fuzz_filter_profile(m, ctx, rng):
    n_groups = int(rng.choice([3, 40, 700, 2500]))
    rec = random_batch(int(rng.integers(1 << 30)), n_groups=n_groups)
    rec.rflags[:] |= samio.HAS_AS
    goff = m.filter_pools(rec)
    batch = m.DeviceBatch.upload(ctx, rec, goff)
    opts = [dict(besthit=True), dict(uniqhit=True), dict(l=15, p=70, besthit=True), dict(z=40, uniqhit=True)][int(rng.integers(4))]
    multi = ["proportional", "equal", "all", "ignore"][int(rng.integers(4))]
    w = orc.run_filter(rec, **opts)
    ref = orc.run_profile(rec, 37, multi=multi, sel=w["emit"])
    run = m.FilterRun(ctx, batch, **opts)
    prof = m.Profile(ctx, 37, multi)
    run.enqueue_with_profile(prof)
    run.finish()
    assert (run.result().emit == w["emit"]).all(), (opts, multi, n_groups)
    assert (prof.ui() == ref["ui"]).all(), (opts, multi, n_groups)
    ab, st = prof.finalize()
    s = ref["stats"]
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
    want = ref["abundance"]
    assert ((ab == 0) == (want == 0)).all()
    assert (np.abs(ab - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all()
    prof.close()
    run.free()
    batch.free()


def op_fuzz_plain_filter(m, ctx, rng):
    rec = random_batch(int(rng.integers(1 << 30)), n_groups=int(rng.choice([5, 300, 2000])))
    batch = m.DeviceBatch.upload(ctx, rec, None)
    got, want = m.aln_stats(ctx, batch), orc.aln_stats(rec)
    for k in ("length", "qlen", "qclip", "edit", "status"):
        assert (got[k] == want[k]).all(), k
    opts = [dict(l=20), dict(p=90), dict(ppt=-930), dict(l=10, p=80, z=50), dict(p=95, invert=True, keep_unmapped=True),
            dict(l=30, rescore=True)][int(rng.integers(6))]
    res, w = m.run_filter(ctx, batch, **opts), orc.run_filter(rec, **opts)
    assert res.n_emit == len(w["emit"]) and (res.emit == w["emit"]).all(), opts
    batch.free()


def op_synth_pipe(m, ctx, rng):
    n_groups, n_refs = int(rng.choice([200, 9000, 70000, 250000])), int(rng.choice([7, 300, 20000]))
    seed = int(rng.integers(1 << 20))
    hs = m.HostSynth(seed, n_groups, n_refs, 4)
    db = m.DeviceBatch.synth(ctx, seed, n_groups, n_refs, 4)
    opts = dict(l=80, p=95, z=80, besthit=True)
    multi = ["proportional", "equal", "all", "ignore"][int(rng.integers(4))]
    w = orc.run_filter(hs, **opts)
    ref = orc.run_profile(hs, n_refs, multi=multi, sel=w["emit"])
    run = m.FilterRun(ctx, db, **opts)
    prof = m.Profile(ctx, n_refs, multi)
    run.enqueue_with_profile(prof)
    run.finish()
    assert (run.result().emit == w["emit"]).all()
    assert (prof.ui() == ref["ui"]).all()
    ab, st = prof.finalize()
    s, want = ref["stats"], ref["abundance"]
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count, st.iterations) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count, s.iterations)
    assert ((ab == 0) == (want == 0)).all()
    assert (np.abs(ab - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all()
    prof.close()
    run.free()
    db.free()


def op_coverage(m, ctx, rng):
    n_groups, n_refs = int(rng.choice([30, 4000, 60000])), int(rng.choice([3, 500, 9000]))
    tl = 5000
    hs = m.HostSynth(int(rng.integers(1 << 20)), n_groups, n_refs, 4)
    if rng.random() < 0.5:
        hs = _cov_fuzz_records(hs, int(rng.integers(1 << 20)), n_refs, tl)
    tlen = [tl] * n_refs
    want = orc.coverage(hs, tlen)
    form = int(rng.integers(4))
    whole = m.DeviceBatch.upload(ctx, m.RecordSlice(hs, 0, hs.n_records))
    try:
        if form == 0:
            got = m.coverage(ctx, whole, tlen)                                  # streamed
        elif form == 1:
            got = m.coverage(ctx, whole, tlen, whole_sample=True)               # one word per run piece
        elif form == 2:
            os.environ["MSX_COV_MARKS"] = "1"
            try:
                got = m.coverage(ctx, whole, tlen, whole_sample=True)           # +1 / -1 marks
            finally:
                del os.environ["MSX_COV_MARKS"]
        else:
            k = int(rng.integers(1, 25))
            edges = np.sort(np.concatenate([[0, hs.n_records], rng.integers(0, hs.n_records + 1, k - 1)])).tolist()
            parts = [m.DeviceBatch.upload(ctx, m.RecordSlice(hs, lo, hi)) for lo, hi in zip(edges[:-1], edges[1:])]
            try:
                got, _ = m.coverage_collected(ctx, parts, tlen)
            finally:
                for p in parts:
                    p.free()
    finally:
        whole.free()
    for t in range(n_refs):
        assert (got[t] == want[t]).all(), (form, t, n_groups, n_refs)


def op_bgzf_round_trip(m, ctx, rng):
    n = int(rng.choice([0, 1, 300, 0xff00, 0xff01, 700_000, 3_000_000]))
    kind = int(rng.integers(3))
    if kind == 0:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()                        # incompressible
    elif kind == 1:
        data = (b"ACGTTGCA" * (n // 8 + 1))[:n]                                           # one long match
    else:
        words = [bytes(rng.integers(65, 91, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(50)]
        data = b"".join(words[int(i)] for i in rng.integers(0, 50, n // 6 + 1))[:n]      # text-like
    level = int(rng.choice([0, 1, 2, 6, 9]))
    stream, n_blk = m.bgzf_deflate(ctx, data, level)
    parts = m.bgzf_split(stream)              # (payload, isize, crc32) per block, every header checked
    assert len(parts) == n_blk
    chunks = [zlib.decompress(pl, -15) for pl, _, _ in parts]
    assert b"".join(chunks) == data
    assert all(len(c) == isize and zlib.crc32(c) & 0xffffffff == crc for c, (_, isize, crc) in zip(chunks, parts))
    # and through the device inflater
    if n_blk:
        comp, blocks, total = m.bgzf_blocks([pl for pl, _, _ in parts], chunks, gap=int(rng.integers(0, 5)))
        out, st, refused = m.bgzf_inflate(ctx, comp, blocks, n_blk, total)
        assert refused == 0 and (st == 0).all() and bytes(out[:total]) == data


OPS = [op_fuzz_filter_profile, op_fuzz_plain_filter, op_synth_pipe, op_coverage, op_bgzf_round_trip]


# MSX_SOAK_SEEDS="100:160" MSX_SOAK_STEPS=30: a longer campaign (profiles/round5/state_soak.log)
_lo, _hi = (int(x) for x in os.environ.get("MSX_SOAK_SEEDS", "11:14").split(":"))


@pytest.mark.parametrize("seed", range(_lo, _hi))
def test_random_sequence_of_operations_on_one_context(ctx, seed):
    import msamtools_amd as m
    rng = np.random.default_rng(seed)
    trace = []
    for step in range(int(os.environ.get("MSX_SOAK_STEPS", "14"))):
        op = OPS[int(rng.integers(len(OPS)))]
        trace.append(op.__name__)
        try:
            op(m, ctx, rng)
        except AssertionError as exc:
            raise AssertionError(f"seed {seed}, step {step}, after {trace}: {exc}") from exc
