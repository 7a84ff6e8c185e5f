"""N ranks emulated on ONE GPU (-m gpu): N contexts, each with the Profile of its own shard, driven through the split calls
a multi-GPU caller uses (msx_profile_accumulators / prop_begin / prop_local / prop_apply / prop_purged) with the
all-reduces done by hand -- the ranks' vectors fetched, added in rank order on the host, written back to every rank --
where msx_profile_finalize_dist_enqueue has RCCL do them (counts once, `share` inside each of the 19 iterations, the purged
count at the end: msx_prop.hip).  What this pins on a one-GPU box is everything about the sharded algorithm but the
collective itself (which the one-rank communicator tests run): shard-local sharing stores whose partial sums add up to the
whole sample's increment (msam_profile.c:341-365 is a sum over ALL multi-mapper lists), the same convergence decision on
every rank, counts and purged inserts that add up -- against one context that accumulated every shard, and the oracle on
the whole stream.  (tests/test_gpu_two_ranks.py is the real thing and needs two GPUs.)"""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
OPTS = dict(l=80, p=95, z=80, besthit=True)


def host_allreduce(ranks, ptrs, count, dtype):
    tot = None
    for (ctx, _), p in zip(ranks, ptrs):
        v = ctx.to_host(p, count, dtype)
        tot = v.copy() if tot is None else tot + v          # rank order: a fixed summation order
    for (ctx, _), p in zip(ranks, ptrs):
        ctx.to_dev(p, tot)
    return tot


@pytest.mark.parametrize("n_ranks", [2, 3, 5])
@pytest.mark.parametrize("groups,refs", [(900, 11), (60_000, 500), (300_000, 20_000)])
@pytest.mark.parametrize("multi,slices", [("proportional", 1), ("proportional", 2), ("proportional", 3), ("equal", 1)])
def test_emulated_ranks_equal_one_context_and_oracle(n_ranks, groups, refs, multi, slices):
    import msamtools_amd as m
    seed = 97531
    per = groups // n_ranks
    bounds = [r * per for r in range(n_ranks)] + [groups]          # (the last rank takes the remainder)
    hs = m.HostSynth(seed, groups, refs, 4)
    ranks, whole_ctx = [], m.Context(0)
    whole = m.Profile(whole_ctx, refs, multi)
    emit_n = 0
    try:
        for r in range(n_ranks):
            ctx = m.Context(0)
            db = m.DeviceBatch.synth(ctx, seed, bounds[r + 1] - bounds[r], refs, 4, first_group=bounds[r])
            prof = m.Profile(ctx, refs, multi)
            run = m.FilterRun(ctx, db, **OPTS)
            run.enqueue_with_profile(prof)
            run.finish()
            emit_n += run.result().n_emit
            run.free()
            # the same shard into the one context that sees everything
            dbw = m.DeviceBatch.synth(whole_ctx, seed, bounds[r + 1] - bounds[r], refs, 4, first_group=bounds[r])
            runw = m.FilterRun(whole_ctx, dbw, **OPTS)
            runw.enqueue_with_profile(whole)
            runw.finish()
            runw.free()
            dbw.free()
            db.free()
            ranks.append((ctx, prof))
        # ---- msx_profile_allreduce_counts by hand: ui, {inserts, uniq, multi}, d (--multi equal) ----
        acc = [p.accumulators() for _, p in ranks]
        host_allreduce(ranks, [a[0] for a in acc], refs, np.uint32)
        host_allreduce(ranks, [a[2] for a in acc], 3, np.uint32)
        if multi == "equal":
            host_allreduce(ranks, [a[1] for a in acc], refs, np.float64)
        for _, p in ranks:
            p.prop_begin()
        iterations, purged = 0, 0
        if multi == "proportional":
            for k in range(1, 20):
                if slices == 1:
                    ptrs = [p.prop_local() for _, p in ranks]
                    host_allreduce(ranks, ptrs, refs, np.float64)
                else:
                    # MSX_DIST_SLICES: the local half slice by slice (msx_profile_prop_local_slice); the ranges are cuts of the
                    # feature range -- the same on every rank, whatever its shard holds -- and a range is all-reduced as soon as
                    # every rank has completed it, while the later slices are still to be computed
                    covered = 0
                    for i in range(slices):
                        got = [p.prop_local_slice(i, slices) for _, p in ranks]
                        assert len({(f, c) for _, f, c in got}) == 1, got          # one range for all ranks
                        _, first, count = got[0]
                        assert first == covered and count >= 0
                        covered += count
                        if count:
                            host_allreduce(ranks, [ptr + 8 * first for ptr, _, _ in got], count, np.float64)
                    assert covered == refs
                deltas = [p.prop_apply() for _, p in ranks]
                iterations = k
                assert len(set(deltas)) == 1, deltas                       # the same numbers, the same decision everywhere
                if deltas[0] < 1e-10:
                    break
            purged = sum(p.prop_purged() for _, p in ranks)
        ab = [ctx.to_host(p.abundance_ptr(), refs, np.float64) for ctx, p in ranks]
        for other in ab[1:]:
            assert np.array_equal(ab[0], other)                            # bit for bit on every rank
        cnt = ranks[0][0].to_host(acc[0][2], 3, np.uint32)
        # ---- one context, every shard ----
        abw, stw = whole.finalize()
        # ---- the oracle on the whole stream ----
        w = orc.run_filter(hs, **OPTS)
        assert emit_n == len(w["emit"])
        ref = orc.run_profile(hs, refs, multi=multi, sel=w["emit"])
        s = ref["stats"]
        assert tuple(int(x) for x in cnt) == (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count) == \
            (stw.insert_count, stw.uniq_mapper_count, stw.multi_mapper_count)
        if multi == "proportional":
            assert (iterations, purged) == (s.iterations, s.purged_insert_count) == (stw.iterations, stw.purged_insert_count)
        want = ref["abundance"]
        assert np.array_equal(ab[0] == 0, want == 0) and np.array_equal(abw == 0, want == 0)
        rel = lambda x, y: float((np.abs(x - y) / np.maximum(np.abs(y), 1e-300)).max())
        assert rel(ab[0], abw) <= 1e-9, rel(ab[0], abw)                    # the same additions in another order
        assert rel(ab[0], want) <= 1e-6, rel(ab[0], want)                  # msam_profile.c:317-410, BASELINE's bound
    finally:
        for ctx, p in ranks:
            p.close()
            ctx.close()
        whole.close()
        whole_ctx.close()


@pytest.mark.parametrize("n_ctx", [2, 3])
@pytest.mark.parametrize("groups,refs", [(60_000, 500), (300_000, 20_000)])
def test_merged_contexts_give_one_contexts_bits_for_multi_equal(n_ctx, groups, refs):
    """One process, several contexts (MSX_DEVICES=0,0,0): every context counts its batches, msx_profile_merge adds them up on the
    first.  --multi equal's shares of 1/k are integers in units of 1/lcm(1..24) until d[] is read (msx_count.h): the merge adds
    the INTEGERS, so the one division per feature sees the same numerator one context would have -- the abundances are
    that context's bit for bit (msam_profile.c:175-182; folding every side before adding rounded N quotients instead of one)."""
    import msamtools_amd as m
    seed = 24680
    per = groups // n_ctx
    bounds = [r * per for r in range(n_ctx)] + [groups]
    parts, whole_ctx = [], m.Context(0)
    whole = m.Profile(whole_ctx, refs, "equal")
    try:
        for r in range(n_ctx):
            ctx = m.Context(0)
            prof = m.Profile(ctx, refs, "equal")
            for c, pr in ((ctx, prof), (whole_ctx, whole)):
                db = m.DeviceBatch.synth(c, seed, bounds[r + 1] - bounds[r], refs, 4, first_group=bounds[r])
                run = m.FilterRun(c, db, **OPTS)
                run.enqueue_with_profile(pr)
                run.finish()
                run.free()
                db.free()
            parts.append((ctx, prof))
        for _, other in parts[1:]:
            parts[0][1].merge(other)
        ab, st = parts[0][1].finalize()
        abw, stw = whole.finalize()
        assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count) == (stw.insert_count, stw.uniq_mapper_count, stw.multi_mapper_count)
        assert np.array_equal(ab, abw)                                     # every bit
    finally:
        for ctx, p in parts:
            p.close()
            ctx.close()
        whole.close()
        whole_ctx.close()
