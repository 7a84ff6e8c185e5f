"""msx_unpack -- the record walk on the device -- against an independent host parse (-m gpu).

The inflated BAM byte stream (gzip-decompressed in Python, header skipped) is fed in chunks cut anywhere; what comes
back per batch -- record offsets, FLAG, tid, pos, the MD/NM/AS presence bits and values, packed CIGAR and MD, the
pools of the chosen loop (msam_filter.c:120-125,170 / msam_profile.c:223-232), the batch's end at a pool boundary,
the bytes carried into the next batch -- is compared with tests/samio.read_bam (struct-based parse) and with
msamtools_amd.grouping (the string rules restated in Python).  msx_unpack_emit is compared with the records' own bytes.
"""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import samio
from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)


@pytest.fixture(scope="module")
def ctx():
    import msamtools_amd as m
    c = m.Context(0)
    yield c
    c.close()


def record_stream(path):
    """(bytes of the records with their block_size prefixes, n_targets)"""
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        p += 8 + l_name
    return raw[p:], n_ref


def offsets_of(stream):
    off, p = [0], 0
    while p < len(stream):
        p += 4 + struct.unpack_from("<i", stream, p)[0]
        off.append(p)
    return np.asarray(off, dtype=np.int64)


def as_blocks(chunk, level=6, piece=65280):
    """the chunk as BGZF payloads: (compressed buffer, table, number of blocks)"""
    import zlib
    import msamtools_amd as m
    datas = [chunk[i:i + piece] for i in range(0, len(chunk), piece)]
    payloads = []
    for d in datas:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payloads.append(co.compress(d) + co.flush())
    comp, blocks, total = m.bgzf_blocks(payloads, datas)
    assert total == len(chunk)
    return comp, blocks, len(datas)


def run_chunks(ctx, stream, n_ref, rec, chunk_sizes, pool_mode, cut_mapped=False, want_stats=True, prefetch=False, feed="plain"):
    """feeds the stream in chunks, checks every batch against `rec`; returns the list of (first record, n, groups).
    feed: "plain" inflated bytes (msx_unpack_enqueue); "bgzf" the chunk as compressed blocks (msx_unpack_enqueue_bgzf);
    "bgzf_ahead" the same with the next chunk's blocks sent and inflated ahead (msx_unpack_prefetch_bgzf)"""
    import msamtools_amd as m
    ahead = []            # (chunk length, handle) of the chunks sent ahead, oldest first: two at most
    pools = {0: None, 1: m.filter_pools(rec), 2: m.profile_pools(rec)}[pool_mode]
    off = offsets_of(stream)
    up = m.Unpack(ctx)
    up.seed()
    first, pos, k, out, fed = 0, 0, 0, [], 0
    try:
        while True:
            sz = chunk_sizes[k % len(chunk_sizes)]
            k += 1
            chunk = stream[pos:pos + sz]
            pos += len(chunk)
            last = pos >= len(stream)
            apos = pos + sum(n for n, _ in ahead[1:]) if ahead else pos     # where the chunks not yet sent ahead begin
            kw = dict(pool_mode=pool_mode, n_targets=n_ref, last=last, cut_mapped=cut_mapped, want_stats=want_stats)
            if feed == "plain" or len(chunk) == 0:
                up.enqueue(chunk, **kw)
            else:
                if ahead:
                    n0, h = ahead.pop(0)
                    assert n0 == len(chunk)
                    up.enqueue_bgzf(h, **kw)
                else:
                    up.enqueue_bgzf(*as_blocks(chunk, level=(1, 6, 9)[k % 3]), **kw)
                if feed == "bgzf_ahead":                  # between enqueue and finish: nothing of it depends on the carry
                    while len(ahead) < 2 and apos < len(stream):
                        j = k + len(ahead)
                        nxt = stream[apos:apos + chunk_sizes[j % len(chunk_sizes)]]
                        ahead.append((len(nxt), up.prefetch_bgzf(*as_blocks(nxt, level=(1, 6, 9)[(j + 1) % 3]))))
                        apos += len(nxt)
            res, view = up.finish()
            if prefetch and not last:        # the next chunk goes up while this batch is looked at
                up.prefetch(stream[pos:pos + chunk_sizes[k % len(chunk_sizes)]])
            fed += len(chunk)
            n = int(res.n_records)
            # the batch covers whole records from where the previous one ended, and the carry is the rest of what was fed
            assert res.bytes_consumed + res.carry_bytes == (fed - off[first])
            assert off[first] + res.bytes_consumed == off[first + n]
            if n:
                sl = slice(first, first + n)
                ro = up.offsets(n)
                assert np.array_equal(ro.astype(np.int64), off[first:first + n + 1] - off[first])
                assert np.array_equal(view.fetch("flag", n, np.uint16), rec.flag[sl])
                assert np.array_equal(view.fetch("tid", n, np.int32), rec.tid[sl])
                assert np.array_equal(view.fetch("pos", n, np.int32), rec.pos[sl])
                assert np.array_equal(view.fetch("rflags", n, np.uint8), rec.rflags[sl])
                assert np.array_equal(view.fetch("nm", n, np.int32), rec.nm[sl])
                assert np.array_equal(view.fetch("as_", n, np.int32), rec.as_[sl])
                if want_stats:
                    co = view.fetch("cigar_off", n + 1, np.uint32).astype(np.int64)
                    assert np.array_equal(co, rec.cigar_off[first:first + n + 1].astype(np.int64) - int(rec.cigar_off[first]))
                    assert np.array_equal(view.fetch("cigar", int(co[-1]), np.uint32),
                                          rec.cigar[int(rec.cigar_off[first]):int(rec.cigar_off[first + n])])
                    mo = view.fetch("md_off", n + 1, np.uint32).astype(np.int64)
                    assert np.array_equal(mo, rec.md_off[first:first + n + 1].astype(np.int64) - int(rec.md_off[first]))
                    assert np.array_equal(view.fetch("md", int(mo[-1]), np.uint8),
                                          rec.md[int(rec.md_off[first]):int(rec.md_off[first + n])])
                if pools is not None:
                    ng = int(res.n_groups)
                    g = view.fetch("group_off", ng + 1, np.uint32).astype(np.int64) + first
                    lo = int(np.searchsorted(pools, first))
                    assert pools[lo] == first, "a batch begins at a pool boundary"
                    assert np.array_equal(g, pools[lo:lo + ng + 1].astype(np.int64)), "the batch's pools are the loop's pools"
                    if not last:
                        assert g[-1] == first + n
                out.append((first, n, int(res.n_groups)))
            first += n
            if last:
                break
        assert first == rec.flag.shape[0]
        assert res.carry_bytes == 0
    finally:
        up.close()
    return out


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    d = tmp_path_factory.mktemp("unpack")
    p = str(d / "s.bam")
    with open(p, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "150000", "--refs", "700", "-u"], stdout=fh)
    stream, n_ref = record_stream(p)
    return stream, n_ref, samio.read_bam(p)[1]


@pytest.mark.parametrize("pool_mode", [0, 1, 2])
def test_synthetic_stream_in_ragged_chunks(ctx, synth, pool_mode):
    stream, n_ref, rec = synth
    batches = run_chunks(ctx, stream, n_ref, rec, [3_000_001, 777_777, 5_000_000, 123_456, 64], pool_mode)
    assert len(batches) >= 8


def test_prefetched_chunks(ctx, synth):
    """msx_unpack_prefetch: the next chunk's bytes sent ahead on the copy stream, behind the carry"""
    stream, n_ref, rec = synth
    batches = run_chunks(ctx, stream, n_ref, rec, [3_000_001, 777_777, 5_000_000, 123_456], 1, prefetch=True)
    assert len(batches) >= 8
    import msamtools_amd as m
    up = m.Unpack(ctx)
    up.seed()
    up.prefetch(stream[:1000])
    with pytest.raises(m.MsxError, match="other bytes"):
        up.enqueue(stream[:2000], pool_mode=1, n_targets=n_ref)
    up.close()


@pytest.mark.parametrize("feed", ["bgzf", "bgzf_ahead"])
@pytest.mark.parametrize("pool_mode", [1, 2])
def test_chunks_that_arrive_compressed(ctx, synth, feed, pool_mode):
    """msx_unpack_enqueue_bgzf / msx_unpack_prefetch_bgzf: the same batches as from inflated bytes"""
    stream, n_ref, rec = synth
    sizes = [3_000_001, 777_777, 5_000_000, 123_456, 64]
    plain = run_chunks(ctx, stream, n_ref, rec, sizes, pool_mode)
    assert run_chunks(ctx, stream, n_ref, rec, sizes, pool_mode, feed=feed) == plain


def test_a_refused_block_leaves_the_carry_alone(ctx, synth):
    """a damaged block: msx_unpack_finish says MSX_ERR_INFLATE and has consumed nothing -- the same chunk handed over
    inflated (what the command line does with such a batch) continues the stream"""
    import msamtools_amd as m
    stream, n_ref, rec = synth
    off = offsets_of(stream)
    up = m.Unpack(ctx)
    up.seed()
    a, b = stream[:2_000_000], stream[2_000_000:4_500_000]
    up.enqueue_bgzf(*as_blocks(a), pool_mode=1, n_targets=n_ref)
    r1, _ = up.finish()
    comp, blocks, nb = as_blocks(b)
    bad = bytearray(comp)
    for q in range(blocks[3].in_off + 40, blocks[3].in_off + 60):
        bad[q] ^= 0x5a
    up.enqueue_bgzf(bytes(bad), blocks, nb, pool_mode=1, n_targets=n_ref)
    with pytest.raises(m.MsxError, match="refused by the device inflater") as ei:
        up.finish()
    assert ei.value.code == -15
    up.enqueue(b, pool_mode=1, n_targets=n_ref)
    r2, view = up.finish()
    n1, n2 = int(r1.n_records), int(r2.n_records)
    assert n1 > 0 and n2 > 0
    assert off[n1] == r1.bytes_consumed and off[n1 + n2] - off[n1] == r2.bytes_consumed
    assert np.array_equal(view.fetch("flag", n2, np.uint16), rec.flag[n1:n1 + n2])
    # sent ahead and damaged: the verdict arrives with the batch it belongs to
    c = stream[4_500_000:6_000_000]
    comp, blocks, nb = as_blocks(c)
    bad = bytearray(comp)
    bad[blocks[1].in_off + 30] ^= 0xff
    h = up.prefetch_bgzf(bytes(bad), blocks, nb)
    up.enqueue_bgzf(h, pool_mode=1, n_targets=n_ref)
    with pytest.raises(m.MsxError, match="refused by the device inflater"):
        up.finish()
    up.enqueue(c, pool_mode=1, n_targets=n_ref)
    r3, view = up.finish()
    n3 = int(r3.n_records)
    assert np.array_equal(view.fetch("tid", n3, np.int32), rec.tid[n1 + n2:n1 + n2 + n3])
    # other blocks than the ones sent ahead
    h = up.prefetch_bgzf(*as_blocks(c[:100_000]))
    with pytest.raises(m.MsxError, match="other blocks"):
        up.enqueue_bgzf(*as_blocks(c[:200_000]), pool_mode=1, n_targets=n_ref)
    up.close()


def test_whole_stream_at_once_and_tiny_chunks(ctx, synth):
    stream, n_ref, rec = synth
    assert len(run_chunks(ctx, stream, n_ref, rec, [len(stream)], 1)) == 1
    sub = offsets_of(stream)[3000]
    import msamtools_amd as m

    class V:
        pass
    v = V()
    for k in ("flag", "tid", "pos", "rflags", "nm", "as_"):
        setattr(v, k, getattr(rec, k)[:3000])
    v.cigar_off, v.md_off, v.cigar, v.md = rec.cigar_off[:3001], rec.md_off[:3001], rec.cigar, rec.md
    v.qname_off, v.qname = rec.qname_off[:3001], rec.qname
    run_chunks(ctx, stream[:sub], n_ref, v, [997, 13, 4096, 1], 1)          # chunks smaller than a record


def odd_sam(tmp_path):
    """aux types and orders the synthetic stream does not have, unmapped records, long names, a B array, a record longer
    than a chase segment"""
    rnd = np.random.RandomState(3)
    lines = ["@HD\tVN:1.6\tSO:queryname"] + [f"@SQ\tSN:c{i}\tLN:900000" for i in range(9)]
    for r in range(6000):
        q = f"read_{r:05d}" + ("x" * int(rnd.randint(0, 200)) if r % 97 == 0 else "")
        for a in range(int(rnd.choice([1, 1, 2, 4]))):
            if rnd.rand() < 0.1:
                lines.append(f"{q}\t{4 | (0x40 if a else 0)}\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\tYT:Z:UU")
                continue
            mm = int(rnd.randint(0, 3))
            aux = [f"NM:i:{int(rnd.choice([mm, 300, 70000]))}", f"MD:Z:{'20A29' if mm else '50'}", f"AS:i:{int(rnd.choice([50 - mm, -3, 200, 40000]))}",
                   "XB:B:s,1,-2,3", "XZ:Z:some text", "XA:A:q", "XF:f:1.5"]
            if rnd.rand() < 0.2:
                aux = [x for x in aux if not x.startswith("MD")]
            if rnd.rand() < 0.1:
                aux = [x for x in aux if not x.startswith("AS")]
            rnd.shuffle(aux)
            seq = "ACGT" * 12 + "AC"
            if r % 1500 == 7 and a == 0:
                seq = "ACGT" * 6000                    # 24 kb of bases: the record spans two chase segments
            cig = f"{len(seq)}M" if len(seq) != 50 else str(rnd.choice(["50M", "10S40M", "20M2D30M", "25M1I24M", "5H45M5S"]))
            lines.append(f"{q}\t{int(rnd.choice([0, 16, 65, 129, 256, 321]))}\tc{int(rnd.randint(0, 9))}\t{int(rnd.randint(1, 800000))}\t60\t{cig}\t*\t0\t0\t"
                         f"{seq if cig.endswith('M') and 'S' not in cig and 'I' not in cig and 'D' not in cig and 'H' not in cig else '*'}\t*\t" + "\t".join(aux))
        if rnd.rand() < 0.15:
            lines.append(f"lonely_{r}\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*")
    p = tmp_path / "odd.sam"
    p.write_text("\n".join(lines) + "\n")
    bam = str(tmp_path / "odd.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", str(p)], stdout=fh)
    return bam


@pytest.mark.parametrize("pool_mode", [1, 2])
def test_odd_records(ctx, tmp_path, pool_mode):
    bam = odd_sam(tmp_path)
    stream, n_ref = record_stream(bam)
    rec = samio.read_bam(bam)[1]
    assert (rec.flag & 4).any() and int(np.diff(offsets_of(stream)).max()) > 16384
    run_chunks(ctx, stream, n_ref, rec, [200_000, 50_001, 1_000_003], pool_mode)
    run_chunks(ctx, stream, n_ref, rec, [len(stream)], pool_mode, cut_mapped=True)


def test_emit_returns_the_records_bytes(ctx, synth):
    import msamtools_amd as m
    stream, n_ref, rec = synth
    off = offsets_of(stream)
    up = m.Unpack(ctx)
    up.seed()
    up.enqueue(stream[:int(off[40000]) + 11], pool_mode=1, n_targets=n_ref, last=False)
    res, view = up.finish()
    n = int(res.n_records)
    assert 0 < n <= 40000
    rnd = np.random.RandomState(1)
    emit = np.sort(rnd.choice(n, size=n // 3, replace=False)).astype(np.int32)
    emit[5:9] = emit[5:9][::-1]                       # any order the caller asks for
    d = ctx.alloc(4 * emit.size)
    ctx.to_dev(d, emit)
    got = up.emit(d, emit.size, len(stream))
    want = b"".join(stream[int(off[i]):int(off[i + 1])] for i in emit)
    assert got == want
    ctx.free(d)
    up.close()


def test_truncated_and_corrupt_streams_are_refused(ctx, synth):
    import msamtools_amd as m
    stream, n_ref, rec = synth
    off = offsets_of(stream)
    up = m.Unpack(ctx)
    up.seed()
    up.enqueue(stream[:int(off[100]) + 7], pool_mode=1, n_targets=n_ref, last=True)
    with pytest.raises(m.MsxError, match="Truncated BAM record"):
        up.finish()
    up.close()
    bad = bytearray(stream[:int(off[100])])
    bad[int(off[50]):int(off[50]) + 4] = struct.pack("<i", 5)            # block_size < 32 on the true chain
    up = m.Unpack(ctx)
    up.seed()
    up.enqueue(bytes(bad), pool_mode=1, n_targets=n_ref, last=True)
    with pytest.raises(m.MsxError, match="Corrupt BAM record"):
        up.finish()
    up.close()


def test_device_unpack_feeds_the_filter(ctx, synth):
    """the unpacked view straight into filter --besthit | profile against the oracle on the host-parsed records"""
    import msamtools_amd as m
    import oracle_lib as orc
    stream, n_ref, rec = synth
    up = m.Unpack(ctx)
    up.seed()
    up.enqueue(stream, pool_mode=1, n_targets=n_ref, last=True)
    res, view = up.finish()
    assert int(res.n_records) == rec.flag.shape[0]
    run = m.FilterRun(ctx, view, l=80, p=95, z=80, besthit=True)
    prof = m.Profile(ctx, n_ref, "proportional")
    run.enqueue_with_profile(prof)
    run.finish()
    got = run.result()
    want = orc.run_filter(rec, l=80, p=95, z=80, besthit=True)
    assert got.emit.tolist() == want["emit"].tolist()
    ab, st = prof.finalize()
    ref = orc.run_profile(rec, n_ref, multi="proportional", sel=want["emit"])
    assert (st.insert_count, st.uniq_mapper_count, st.multi_mapper_count) == \
        (ref["stats"].insert_count, ref["stats"].uniq_mapper_count, ref["stats"].multi_mapper_count)
    assert (np.abs(ab - ref["abundance"]) / np.maximum(np.abs(ref["abundance"]), 1e-300)).max() <= 1e-6
    prof.close()
    run.free()
    up.close()
