"""`filter ... | profile -` in one call against the oracle's pipe on streams where filter's pools and profile's pools
differ (-m gpu).

The filter loop closes a pool at every record whose QNAME differs from the last MAPPED record's
(msam_filter.c:120-125,170); profile re-pools filter's output by QNAME (msam_profile.c:223-232).  A mapped, B
unmapped, A mapped is two filter pools and -- B is never written -- one insert.  The device counts a pool that
begins with an unmapped record together with the pool before it (msx_batch.pool_rule = MSX_POOLS_FILTER,
msx_count.h).  Three forms are compared with the oracle run as the reference's two commands (orc.run_filter, then
orc.run_profile over the emitted records, pooled by QNAME string):
  fused     msx_filter_profile_enqueue (the insert accounting inside the best-hit kernel)
  two-call  msx_filter_enqueue, then msx_profile_accumulate(keep)
  plain     the same with a -l/-p/-z filter only (no best hit: pools do not shape filter's output)
"""
import numpy as np
import pytest

import oracle_lib as orc
import samio

pytestmark = pytest.mark.gpu
N_REF = 37


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    c = m.Context(0)
    yield c
    c.close()


def sam_text(records):
    """records: (qname, flag, ref index or None, AS, mismatches)"""
    lines = ["@HD\tVN:1.6\tSO:queryname"] + [f"@SQ\tSN:r{i}\tLN:5000" for i in range(N_REF)]
    for q, flag, ref, score, mm in records:
        if flag & 4:
            rn = "*" if ref is None else f"r{ref}"
            lines.append(f"{q}\t{flag}\t{rn}\t{0 if ref is None else 7}\t0\t*\t*\t0\t0\t*\t*")
        else:
            # 10 matches, then mm times [one mismatch + 9 matches], the last run taking the rest up to 50 bases
            md = "50" if mm == 0 else "10" + "A9" * (mm - 1) + "A" + str(49 - 10 * mm)
            lines.append(f"{q}\t{flag}\tr{ref}\t{100 + ref}\t60\t50M\t*\t0\t0\t*\t*\tNM:i:{mm}\tMD:Z:{md}\tAS:i:{score}")
    return "\n".join(lines) + "\n"


def load(tmp_path, records, name="x.sam"):
    p = tmp_path / name
    p.write_text(sam_text(records))
    return samio.read_sam(str(p))[1]


def oracle_pipe(rec, multi, **opts):
    f = orc.run_filter(rec, **opts)
    assert f["rc"] == 0
    return f, orc.run_profile(rec, N_REF, multi=multi, sel=f["emit"])


def check(st, ui, ab, ref):
    s = ref["stats"]
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count, st.purged_insert_count) == \
        (s.insert_count, s.uniq_mapper_count, s.multi_mapper_count, s.purged_insert_count)
    assert np.array_equal(ui, ref["ui"])
    want = ref["abundance"]
    assert np.array_equal(ab == 0, want == 0)
    assert (np.abs(ab - want) / np.maximum(np.abs(want), 1e-300)).max() <= 1e-6      # north_star tolerance


def gpu_forms(ctx, rec, multi, **opts):
    """[(form, stats, ui, abundance, emit)] for the fused and the two-call form"""
    import msamtools_amd as m
    best = opts.get("besthit") or opts.get("uniqhit")
    # without best hit the pools do not shape filter's output and the host hands over profile's pools
    # (transparent: tid == -1, and every unmapped record -- none is written without -k -v)
    goff = m.filter_pools(rec) if best else profile_pools_of_mapped(rec)
    out = []
    for form in ("fused", "two-call"):
        batch = m.DeviceBatch.upload(ctx, rec, goff, filter_pools=bool(best))
        run = m.FilterRun(ctx, batch, **opts)
        prof = m.Profile(ctx, N_REF, multi)
        try:
            if form == "fused":
                run.enqueue_with_profile(prof)
                run.finish()
            else:
                run.enqueue()
                run.finish()
                prof.accumulate(batch, run.keep)
            res = run.result()
            ui = prof.ui()
            ab, st = prof.finalize()
            out.append((form, st, ui, ab.copy(), res.emit.copy()))
        finally:
            prof.close()
            run.free()
            batch.free()
    return out


def profile_pools_of_mapped(rec):
    """msam_profile.c:223-232 over the records filter can write without -k -v: the mapped ones with a reference."""
    import msamtools_amd as m

    class V:
        pass
    v = V()
    v.qname_off, v.qname, v.flag = rec.qname_off, rec.qname, rec.flag
    v.tid = np.where((rec.flag & 4) != 0, -1, rec.tid).astype(np.int32)
    return m.profile_pools(v)


CASES = {
    # the judge's stream: A mapped, B unmapped, A mapped -> one insert, multi-mapped to r1 and r2
    "a_b_a": [("A", 0, 1, 50, 0), ("B", 4, None, 0, 0), ("A", 256, 2, 50, 0), ("C", 0, 3, 50, 0)],
    # the same reference twice: one insert, unique
    "a_b_a_same_ref": [("A", 0, 1, 50, 0), ("B", 4, None, 0, 0), ("A", 256, 1, 50, 0)],
    # two unmapped names in between (three filter pools, the middle one empty), unmapped with a mate's position
    "a_b_c_a": [("A", 0, 1, 50, 0), ("B", 4, 5, 0, 0), ("C", 4, None, 0, 0), ("A", 256, 2, 50, 0), ("A", 256, 4, 50, 0)],
    # the interleaved record shares the name of the NEXT read: D unmapped, D mapped
    "a_d_a_d": [("A", 0, 1, 50, 0), ("D", 4, None, 0, 0), ("A", 256, 2, 50, 0), ("D", 0, 3, 50, 0)],
    # the stream begins with unmapped records; the second half of A loses the best-hit contest on its own
    "lead_unmapped": [("U", 4, None, 0, 0), ("V", 4, None, 0, 0), ("A", 0, 1, 48, 1), ("W", 4, None, 0, 0),
                      ("A", 256, 2, 50, 0), ("A", 256, 3, 50, 0)],
    # mates: READ1 in the first pool, READ2 behind the unmapped record
    "mates_split": [("A", 65, 1, 50, 0), ("A", 321, 2, 50, 0), ("B", 4, None, 0, 0), ("A", 129, 1, 50, 0), ("A", 385, 6, 50, 0),
                    ("E", 65, 7, 50, 0), ("E", 129, 7, 50, 0)],
    # a mapped read whose every alignment fails the filter between two others: NOT a chain (names differ)
    "failing_read_between": [("A", 0, 1, 50, 0), ("F", 0, 2, 30, 4), ("G", 0, 3, 50, 0)],
}


@pytest.mark.parametrize("multi", ["proportional", "all", "equal", "ignore"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_named_streams(ctx, tmp_path, name, multi):
    rec = load(tmp_path, CASES[name])
    opts = dict(l=30, p=95, z=80, besthit=True)
    f, ref = oracle_pipe(rec, multi, **opts)
    for form, st, ui, ab, emit in gpu_forms(ctx, rec, multi, **opts):
        assert emit.tolist() == f["emit"].tolist(), form
        check(st, ui, ab, ref)
    if name == "a_b_a":
        assert (ref["stats"].insert_count, ref["stats"].multi_mapper_count) == (2, 1)       # A once, C once


def random_stream(seed, n_reads, long_pools=False):
    rnd = np.random.RandomState(seed)
    out = []
    u = 0
    for r in range(n_reads):
        q = f"read{r:06d}"
        k = int(rnd.choice([1, 1, 2, 3, 5])) if not long_pools or rnd.rand() > 0.1 else int(rnd.randint(33, 70))
        paired = rnd.rand() < 0.5
        fam = int(rnd.randint(0, N_REF))
        for a in range(k):
            if rnd.rand() < 0.25:                      # a record of another name, unmapped, inside the read's run
                for _ in range(int(rnd.choice([1, 1, 2]))):
                    out.append((f"un{u:06d}", 4, int(rnd.randint(0, N_REF)) if rnd.rand() < 0.3 else None, 0, 0))
                    u += 1
            mm = int(rnd.choice([0, 0, 0, 1, 2, 4]))
            flag = (0x41 if rnd.rand() < 0.5 else 0x81) if paired else 0
            if a:
                flag |= 0x100
            ref = (fam + int(rnd.choice([0, 0, 1, 2]))) % N_REF
            out.append((q, flag, ref, 50 - 2 * mm - int(rnd.choice([0, 0, 0, 1])), mm))
        if rnd.rand() < 0.2:
            out.append((f"un{u:06d}", 4, None, 0, 0))
            u += 1
    return out


@pytest.mark.parametrize("multi", ["proportional", "equal"])
@pytest.mark.parametrize("seed,n_reads,long_pools", [(1, 3000, False), (2, 3000, True), (3, 40000, False)])
def test_random_streams_besthit(ctx, tmp_path, seed, n_reads, long_pools, multi):
    """Thousands of chains, at every place of the 64-pool tiles of the best-hit kernel (its flat and its
    lane-per-pool path: pools of more than 32 records), --besthit and --uniqhit."""
    rec = load(tmp_path, random_stream(seed, n_reads, long_pools))
    for opts in (dict(l=30, p=95, z=80, besthit=True), dict(uniqhit=True), dict(besthit=True)):
        f, ref = oracle_pipe(rec, multi, **opts)
        plain = orc.run_profile(rec, N_REF, multi=multi)
        for form, st, ui, ab, emit in gpu_forms(ctx, rec, multi, **opts):
            assert emit.tolist() == f["emit"].tolist(), (form, opts)
            check(st, ui, ab, ref)
    # the streams do contain chains: counting one insert per filter pool would differ
    import msamtools_amd as m
    goff = m.filter_pools(rec)
    first_unmapped = (rec.flag[goff[1:-1]] & 4) != 0
    assert first_unmapped.sum() > n_reads // 10


@pytest.mark.parametrize("multi", ["proportional", "all"])
def test_random_streams_plain_filter(ctx, tmp_path, multi):
    """-l/-p/-z only: filter writes records in input order whatever the pools; the host hands over profile's pools."""
    rec = load(tmp_path, random_stream(4, 5000))
    opts = dict(l=30, p=95, z=80)
    f, ref = oracle_pipe(rec, multi, **opts)
    for form, st, ui, ab, emit in gpu_forms(ctx, rec, multi, **opts):
        assert emit.tolist() == f["emit"].tolist(), form
        check(st, ui, ab, ref)


def test_pool_rule_needs_flag(ctx, tmp_path):
    import msamtools_amd as m
    rec = load(tmp_path, CASES["a_b_a"])
    goff = m.filter_pools(rec)

    class NoFlag:
        pass
    v = NoFlag()
    for k in ("rflags", "tid", "pos", "cigar_off", "cigar", "md_off", "md", "nm", "as_"):
        setattr(v, k, getattr(rec, k))
    v.flag = rec.flag
    batch = m.DeviceBatch.upload(ctx, v, goff, filter_pools=True)
    saved, batch.b.flag = batch.b.flag, None
    prof = m.Profile(ctx, N_REF, "proportional")
    with pytest.raises(m.MsxError):
        prof.accumulate(batch, None)
    batch.b.flag = saved
    prof.close()
    batch.free()


def test_tee_cli_counts_chains_across_batch_cuts(ctx, tmp_path):
    """The one-process pipe of the command line (`filter --besthit --profile-out`) on a stream full of interleaved
    unmapped names, cut into many batches: batch ends are placed in front of pools that begin with a mapped record, so
    no chain is cut; header counts and values are the oracle's for the two commands."""
    import gzip
    import os
    import subprocess
    from conftest import ROOT
    BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
    DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)
    recs = random_stream(5, 120000)
    sam = tmp_path / "s.sam"
    sam.write_text(sam_text(recs))
    rec = samio.read_sam(str(sam))[1]
    bam = str(tmp_path / "s.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-u", str(sam)], stdout=fh)
    opts = dict(l=30, p=95, z=80, besthit=True)
    f, ref = oracle_pipe(rec, "proportional", **opts)
    p = str(tmp_path / "p.gz")
    env = dict(os.environ, MSX_BATCH_BYTES="1500000", MSX_BATCH_RECORDS="110000", MSX_THREADS="8", MSX_TIMING="1", MSX_INFLATE_BLOCKS="8")
    r = subprocess.run(f"{BIN} filter -l 30 -p 95 -z 80 --besthit -bu --profile-out {p} --label S {bam} > {tmp_path / 'f.bam'}",
                       shell=True, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    nb = [int(l.split()[2]) for l in r.stderr.decode().split("\n") if l.startswith("# batches:")]
    assert nb and nb[0] >= 3
    out = subprocess.check_output([DEV, "recode", str(tmp_path / "f.bam")]).decode().split("\n")[:-1]
    src = subprocess.check_output([DEV, "recode", bam]).decode().split("\n")[:-1]
    assert out == [src[i] for i in f["emit"]]
    text = gzip.open(p, "rt").read()
    head = [l for l in text.split("\n") if l.startswith("#")]
    get = lambda key: next(l for l in head if l.startswith(key)).split(":")[1].split("(")[0].strip()
    s = ref["stats"]
    assert (int(get("# Mapped inserts")), int(get("#   - Multiple mapped")), int(get("#   - Uniquely mapped"))) == \
        (s.insert_count, s.multi_mapper_count, s.uniq_mapper_count)
    vals, _, _ = orc.profile_finish(ref["abundance"], np.full(N_REF, 5000, np.uint32), s, unit="rel")
    got = np.array([float(l.split("\t")[1]) for l in text.split("\n") if l and not l.startswith("#") and not l.startswith("ID")])
    assert np.allclose(got, vals, rtol=1.1e-6, atol=0)
    # the two-process pipe on the same file agrees (profile re-pools by QNAME itself)
    p2 = str(tmp_path / "p2.gz")
    r = subprocess.run(f"{BIN} filter -l 30 -p 95 -z 80 --besthit -bu {bam} | {BIN} profile --label S -o {p2} -", shell=True,
                       env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0
    strip = lambda t: "\n".join(l for l in t.split("\n") if not l.startswith("# Command"))
    assert strip(gzip.open(p2, "rt").read()) == strip(text)
