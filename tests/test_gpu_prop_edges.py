"""Edge cases of the proportional-sharing build on the device against the oracle: merge weights too large
for the entry key (general path), duplicate lists of more than three features (hashed signatures,
compared entry by entry), feature ids beyond the 21-bit signature fields, one or two features only,
and both ways of counting unique inserts."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu


class Rec:
    """Profile-only records: pools given by name_id, one tid per record."""

    def __init__(self, pools):
        tids = [t for p in pools for t in p]
        n = len(tids)
        self.tid = np.array(tids, dtype=np.int32)
        self.name_id = np.repeat(np.arange(len(pools), dtype=np.int32), [len(p) for p in pools])
        self.group_off = np.concatenate([[0], np.cumsum([len(p) for p in pools])]).astype(np.uint32)
        self.flag = np.zeros(n, np.uint16)
        self.rflags = np.zeros(n, np.uint8)
        self.pos = np.zeros(n, np.int32)
        self.cigar_off = np.zeros(n + 1, np.uint32)
        self.cigar = np.zeros(1, np.uint32)
        self.md_off = np.zeros(n + 1, np.uint32)
        self.md = np.zeros(1, np.uint8)
        self.nm = np.zeros(n, np.int32)
        self.as_ = np.zeros(n, np.int32)
        self.qname_off = None
        self.qname = None


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    c = m.Context(0)
    yield c
    c.close()


def check(ctx, pools, nf, multi="proportional"):
    import msamtools_amd as m
    rec = Rec(pools)
    db = m.DeviceBatch.upload(ctx, rec, rec.group_off)
    prof = m.Profile(ctx, nf, multi)
    prof.accumulate(db, None)
    ui = prof.ui()
    ab, st = prof.finalize()
    ref = orc.run_profile(rec, nf, multi=multi, name_id=rec.name_id)
    s = ref["stats"]
    assert (ui == ref["ui"]).all()
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count, st.iterations) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count, s.iterations)
    want = ref["abundance"]
    assert ((ab == 0) == (want == 0)).all()
    assert (np.abs(ab - want) <= 1e-6 * np.maximum(np.abs(want), 1e-300)).all()
    shared = prof.shared_size()
    prof.close()
    db.free()
    return shared


def mixed_pools(rng, base=0):
    pools = [[base + 5, base + 9]] * 6000                     # one pair 6000 times: weight beyond 12 bits at 2^20 features
    pools += [[base + 5, base + 9, base + 11]] * 3000
    pools += [[base + 1, base + 2, base + 3, base + 4, base + 5]] * 50    # > 3 features: hashed signature, duplicates
    pools += [[base + 1, base + 2, base + 3, base + 4, base + 6]] * 40
    pools += [[base + 7, base + 7, base + 8]] * 30            # repeated feature inside a pool
    pools += [[int(t)] for t in base + rng.integers(0, 20, 20000)]   # unique inserts
    pools += [[-1], [-1, base + 3]]                           # no reference / partly without
    order = rng.permutation(len(pools))
    return [pools[i] for i in order]


@pytest.mark.parametrize("by_part", ["0", "1"])
def test_heavy_weights_and_long_duplicate_lists(ctx, monkeypatch, by_part):
    monkeypatch.setenv("MSX_COUNT_BY_PARTITION", by_part)
    rng = np.random.default_rng(1)
    lists, entries = check(ctx, mixed_pools(rng), 1 << 20)
    assert lists == 5 and entries == 2 + 3 + 5 + 5 + 2          # every distinct set once


def test_feature_ids_beyond_the_signature_fields(ctx, monkeypatch):
    monkeypatch.setenv("MSX_COUNT_BY_PARTITION", "0")           # 2.5 M features: more than the partition count takes
    rng = np.random.default_rng(2)
    lists, entries = check(ctx, mixed_pools(rng, base=2_400_000), 2_500_000)
    assert lists == 5


@pytest.mark.parametrize("nf", [1, 2, 3])
def test_one_two_three_features(ctx, nf):
    rng = np.random.default_rng(nf)
    pools = [[int(t) for t in rng.integers(0, nf, int(rng.integers(1, 4)))] for _ in range(3000)]
    check(ctx, pools, nf)


@pytest.mark.parametrize("multi", ["all", "equal", "ignore"])
def test_other_modes_on_the_mixed_pools(ctx, multi):
    rng = np.random.default_rng(3)
    check(ctx, mixed_pools(rng), 4096, multi=multi)


def test_convergence_decision_next_to_the_threshold(ctx):
    """msam_profile.c:383 stops when DELTA^2 = sum(diff^2) / n_features < 1e-10.  The device adds the same
    numbers in another order (and a * sum(w/S) instead of sum(a/S)), so its DELTA^2 differs from the
    reference's in the last bits -- about 1e-26 absolute here.  Policy: the decision must be the
    reference's whenever DELTA^2 is farther than that from 1e-10.  n_features only divides the sum
    (references without inserts add nothing to it), so the number of features F* at which the decision
    of one iteration flips is found with the oracle by bisection: at F* the loop stops at iteration k,
    at F* - 1 it runs one more, and the two DELTA^2 values straddle 1e-10 by less than 1e-13."""
    import msamtools_amd as m
    refs = 3000
    hs = m.HostSynth(97531, 200000, refs, 1)     # converges at iteration 16 with n_features = refs
    goff = m.profile_pools(hs)

    def orc_iters(nf):
        st = orc.run_profile(hs, nf, multi="proportional")["stats"]
        return st.iterations, st.converged, st.last_delta

    k0, conv0, _ = orc_iters(refs)
    assert conv0 and 3 <= k0 <= 19
    lo, hi = refs, 64 * refs                    # iterations(lo) == k0, iterations(hi) < k0
    assert orc_iters(hi)[0] < k0
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if orc_iters(mid)[0] < k0:
            hi = mid
        else:
            lo = mid
    f_star = hi
    (k_a, c_a, d_a), (k_b, c_b, d_b) = orc_iters(f_star), orc_iters(f_star - 1)
    assert (k_a, c_a) == (k0 - 1, 1) and (k_b, c_b) == (k0, 1)
    # DELTA^2 of iteration k0 - 1 on either side of the threshold: d_a below it, N / (F* - 1) at or above it
    above = d_a * f_star / (f_star - 1)
    assert d_a < 1e-10 <= above and above - d_a < 1e-13
    batch = m.DeviceBatch.upload(ctx, hs, goff)
    try:
        for nf, want_k, want_d in ((f_star, k_a, d_a), (f_star - 1, k_b, d_b)):
            prof = m.Profile(ctx, nf, "proportional")
            prof.accumulate(batch, None)
            ab, st = prof.finalize()
            assert (st.iterations, st.converged) == (want_k, 1), (nf, st.iterations, want_k)
            assert abs(st.delta[want_k] - want_d) <= 1e-9 * want_d
            ref = orc.run_profile(hs, nf, multi="proportional")
            assert (np.abs(ab - ref["abundance"]) <= 1e-6 * np.maximum(np.abs(ref["abundance"]), 1e-300)).all()
            prof.close()
    finally:
        batch.free()


@pytest.mark.parametrize("groups,refs", [(200_000, 50), (200_000, 500), (1_000_000, 5000)])
def test_merged_store_stays_close_to_the_distinct_sets(groups, refs):
    """identical multi-mapper lists are merged only when the sort leaves them adjacent -- the key holds
    the smallest feature and 12 hash bits, so two different sets that collide on it interleave (A, B, A) and some
    merges are missed.  Results do not depend on it (every surviving list keeps its weight), the size of the derived
    store and with it the cost of an iteration does.  Measured on MI355X: 1.03-1.17 x the number of distinct sets
    (scripts/archive/dbg_merge.py); guarded here at 1.25 x, and the sum of the weights must be the number of lists."""
    import msamtools_amd as m
    ctx = m.Context(0)
    db = m.DeviceBatch.synth(ctx, 13579, groups, refs, 4)
    prof = m.Profile(ctx, refs, "proportional")
    try:
        prof.accumulate(db, None)
        prof.finalize_enqueue()
        prof.fetch()
        l0, e0 = prof.multi_size()
        l1, e1 = prof.shared_size()
        hs = m.HostSynth(13579, groups, refs, 4)
        goff = hs.group_off.astype(np.int64)
        sets = set()
        n_multi = 0
        for g in range(groups):
            u = frozenset(hs.tid[goff[g]:goff[g + 1]].tolist())
            if len(u) > 1:
                sets.add(u)
                n_multi += 1
        assert l0 == n_multi
        assert len(sets) <= l1 <= 1.25 * len(sets), (l1, len(sets))
        assert e1 < e0
    finally:
        prof.close()
        db.free()
        ctx.close()
