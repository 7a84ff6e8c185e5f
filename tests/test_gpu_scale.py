"""Full-size GPU checks (BASELINE.json configs[1]: 10 M alignments, 10 k refs,
~5 hits/read) through size-independent properties, plus exact comparison of a
prefix of the big batch with the oracle (pools are independent, so the first
N pools of the big run must equal a run over those N pools alone)."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

N_GROUPS = 2_000_000     # ~10 M alignments
N_REFS = 10_000
PREFIX = 100_000         # pools compared record by record with the oracle
OPTS = dict(l=80, p=95, z=80, besthit=True)


@pytest.fixture(scope="module")
def big():
    import msamtools_amd as m
    ctx = m.Context(0)
    db = m.DeviceBatch.synth(ctx, 13579, N_GROUPS, N_REFS, 4)
    run = m.FilterRun(ctx, db, **OPTS)
    run.enqueue()
    run.finish()
    res = run.result()
    yield ctx, db, run, res
    run.free()
    db.free()
    ctx.close()


def test_c2_shape(big):
    ctx, db, run, res = big
    assert db.n_groups == N_GROUPS and 9_500_000 < db.n_records < 10_500_000


def test_c2_emit_properties(big):
    ctx, db, run, res = big
    goff = db.fetch("group_off", N_GROUPS + 1, np.uint32).astype(np.int64)
    keep, emit = res.keep, res.emit.astype(np.int64)
    assert res.n_emit == int((keep != 0).sum()) == emit.size
    # every record emitted exactly once, and only kept records
    assert np.unique(emit).size == emit.size and (keep[emit] != 0).all()
    # output is pool by pool: the pool id of the emitted records never decreases
    gid = np.searchsorted(goff, emit, side="right") - 1
    assert (np.diff(gid) >= 0).all()
    # within a pool: all pass-1 (READ1) records before pass-2 (READ2), each in input order
    same = np.diff(gid) == 0
    k = keep[emit].astype(np.int64)
    assert (np.diff(k)[same] >= 0).all()
    asc = np.diff(emit) > 0
    assert asc[same & (np.diff(k) == 0)].all()
    # best-hit: every kept record has the maximum AS among its pool's kept-or-not candidates of the same mate
    as_ = db.fetch("as_", db.n_records, np.int32)
    flag = db.fetch("flag", db.n_records, np.uint16)
    mate2 = (flag[emit] & 0x80) != 0
    key = gid * 2 + mate2
    best = {}
    order = np.argsort(key, kind="stable")
    ks, av = key[order], as_[emit][order]
    starts = np.flatnonzero(np.r_[True, np.diff(ks) != 0])
    mx = np.maximum.reduceat(av, starts)
    mn = np.minimum.reduceat(av, starts)
    assert (mx == mn).all()          # all winners of a (pool, mate) tie at the same score


def test_c2_prefix_equals_oracle(big):
    import msamtools_amd as m
    ctx, db, run, res = big
    hs = m.HostSynth(13579, PREFIX, N_REFS, 4)
    want = orc.run_filter(hs, **OPTS)
    n = hs.n_records
    kept = np.zeros(n, bool)
    kept[want["emit"]] = True
    assert ((res.keep[:n] != 0) == kept).all()
    ne = len(want["emit"])
    assert (res.emit[:ne] == want["emit"]).all()


def test_c2_fused_profile_properties(big):
    import msamtools_amd as m
    ctx, db, run, res = big
    prof = m.Profile(ctx, N_REFS, "proportional")
    prof.accumulate(db, run.keep)
    ui = prof.ui().astype(np.int64)
    ab, st = prof.finalize()
    # integer accounting closes: every insert is unique or multi; ui holds 2 per unique insert
    assert st.insert_count == st.uniq_mapper_count + st.multi_mapper_count
    assert ui.sum() == 2 * st.uniq_mapper_count
    kept_groups = np.unique(np.searchsorted(db.fetch("group_off", N_GROUPS + 1, np.uint32).astype(np.int64),
                                            res.emit.astype(np.int64), side="right") - 1).size
    assert st.insert_count == kept_groups
    # mass conservation of proportional sharing: every non-purged multi-mapper adds exactly 1
    total = st.uniq_mapper_count + st.multi_mapper_count - st.purged_insert_count
    assert abs(ab.sum() - total) <= 1e-9 * total
    assert (ab >= ui / 2 - 1e-12).all() and 1 <= st.iterations <= 19
    # idempotence: a second finalize of the same counts gives the same vector within fp-atomic noise
    ab2, st2 = prof.finalize()
    assert st2.iterations == st.iterations and np.allclose(ab, ab2, rtol=1e-9, atol=0)
    prof.close()


def test_bench_multi_gpu_step_on_one_rank():
    """bench.py's N>1 step (torch tensors aliasing the library's device buffers, RCCL all-reduce between
    the split proportional-sharing calls) run with a one-rank process group must reproduce the N=1 result."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--groups", "200000", "--refs", "5000",
            "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--print-checksum"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    a = json.loads(subprocess.check_output(base, env=env).decode().strip().split("\n")[-1])
    b = json.loads(subprocess.check_output(base + ["--force-dist"], env=env).decode().strip().split("\n")[-1])
    assert b["config"]["prop_iterations"] == a["config"]["prop_iterations"]
    ca, cb = a["checksum"], b["checksum"]
    for k in ("inserts", "uniq", "multi", "purged", "iterations"):
        assert ca[k] == cb[k], k
    assert abs(ca["abundance_sum"] - cb["abundance_sum"]) <= 1e-9 * ca["abundance_sum"]
    assert ca["abundance_sha1_6dp"] == cb["abundance_sha1_6dp"]
    # the self-check the N-rank line carries: the distributed result against one context accumulating every shard
    assert b["dist_parity"]["ok"] is True and b["dist_parity"]["max_rel_diff"] <= 1e-9, b["dist_parity"]
    assert "dist_parity" not in a


@pytest.mark.parametrize("groups,refs", [(300, 7), (2000, 50), (60_000, 500), (400_000, 20_000)])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("poll,slices", [("0", "1"), ("8", "1"), ("1", "1"), ("8", "2"), ("0", "3")])
def test_one_rank_distributed_finalize_equals_plain_and_oracle(groups, refs, fused, poll, slices, monkeypatch):
    """msx_profile_finalize_dist_enqueue over a one-rank RCCL communicator -- counts all-reduced, the partial slots of
    k_share_reduce folded into share[] (k_partial_reduce, run by run, no atomics), share[] all-reduced inside each of
    the 19 iterations -- against msx_profile_finalize_enqueue on the same inserts (equal to 1e-12: the same additions
    in another order) and against the oracle (<= 1e-6, msam_profile.c:317-410).  Small inputs matter: most waves of
    k_share_reduce idle then and every feature's segment ends at a chunk's last entry somewhere.
    poll: MSX_DIST_POLL -- never look at the convergence flag (all 19 iterations enqueued), every 8th iteration (the
    default), every iteration: the same numbers, the same iteration count either way.
    slices: MSX_DIST_SLICES -- the local half in slices of the feature range, each slice's all-reduce on the communicator's side
    stream under the next slice's kernels: the same additions in the same order, so every bit of the one-slice form (checked
    below against a second distributed run with one slice)."""
    import msamtools_amd as m
    monkeypatch.setenv("MSX_DIST_POLL", poll)
    monkeypatch.setenv("MSX_DIST_SLICES", slices)
    ctx = m.Context(0)
    ctx.dist_init(m.dist_unique_id(), 0, 1)
    db = m.DeviceBatch.synth(ctx, 24680, groups, refs, 4)
    hs = m.HostSynth(24680, groups, refs, 4)
    out = {}
    try:
        for path in ("plain", "dist") + (("dist1",) if slices != "1" else ()):
            if path == "dist1":
                monkeypatch.setenv("MSX_DIST_SLICES", "1")
            prof = m.Profile(ctx, refs, "proportional")
            if fused:
                run = m.FilterRun(ctx, db, **OPTS)
                run.enqueue_with_profile(prof)
                run.finish()
                sel = run.result().emit
                run.free()
            else:
                prof.accumulate(db, None)
                sel = None
            if path != "plain":
                prof.finalize_dist_enqueue()
            else:
                prof.finalize_enqueue()
            ab, st = prof.fetch()
            out[path] = (ab.copy(), (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count,
                                     st.iterations, st.converged))
            prof.close()
        ref = orc.run_profile(hs, refs, multi="proportional", sel=sel)
    finally:
        db.free()
        ctx.close()
    a, b = out["plain"][0], out["dist"][0]
    assert out["plain"][1] == out["dist"][1]
    if "dist1" in out:
        assert np.array_equal(out["dist1"][0], b) and out["dist1"][1] == out["dist"][1]
    assert np.array_equal(a == 0, b == 0)
    assert (np.abs(a - b) / np.maximum(np.abs(a), 1e-300)).max() <= 1e-12
    s = ref["stats"]
    assert out["dist"][1] == (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count,
                              s.iterations, s.converged)
    want = ref["abundance"]
    assert np.array_equal(b == 0, want == 0)
    assert (np.abs(b - want) / np.maximum(np.abs(want), 1e-300)).max() <= 1e-6
