"""BGZF blocks written on the device (msx_deflate.hip): the reference writes through htslib's BGZF layer
(sam_write1 -> bgzf_write, msam_helper.c:270-272; "wbu" / "wb": msam_filter.c:464-470).  Which bytes a writer puts into
its blocks is not pinned by the reference (its tests compare records: tests/functions.sh:160-163); the rule is that
every block is a valid gzip member with a BC field, that zlib and this repository's own inflater decode the blocks to
the input, byte for byte, and that CRC-32 / ISIZE / BSIZE are right."""
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAYLOAD = 0xff00


def M():
    import msamtools_amd
    return msamtools_amd


@pytest.fixture(scope="module")
def ctx():
    c = M().Context(0)
    yield c
    c.close()


def bam_like(rng, n_bytes, seq=False):
    """records that repeat their neighbours, the way name-sorted BAM does"""
    out = bytearray()
    k = 0
    while len(out) < n_bytes:
        name = b"sim%08d" % k
        for h in range(int(rng.integers(1, 9))):
            core = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
            out += (60 + len(name)).to_bytes(4, "little") + core + name + b"\0" + bytes([100 << 4 & 255, 6, 0, 0])
            if seq:
                out += rng.integers(0, 256, 50, dtype=np.uint8).tobytes() + bytes(rng.integers(30, 42, 100, dtype=np.uint8))
            out += b"NMC" + bytes([int(rng.integers(0, 4))]) + b"ASC" + bytes([int(rng.integers(90, 101))]) + b"MDZ100\0"
        k += 1
    return bytes(out[:n_bytes])


def check_stream(ctx, data, stream, n_blk, level):
    m = M()
    blocks = m.bgzf_split(stream)
    assert len(blocks) == n_blk == (len(data) + PAYLOAD - 1) // PAYLOAD
    # whole-stream check by an independent reader: a BGZF file is a multi-member gzip file
    assert gzip.decompress(stream) == data if data else stream == b""
    pos = 0
    payloads, datas = [], []
    for i, (pl, isize, crc) in enumerate(blocks):
        want = data[pos:pos + isize]
        assert isize == (PAYLOAD if i + 1 < len(blocks) else len(data) - pos)
        assert crc == zlib.crc32(want)
        assert zlib.decompress(pl, -15) == want
        if level == 0:
            assert len(pl) == 5 + isize and pl[0] == 1
        payloads.append(pl)
        datas.append(want)
        pos += isize
    assert pos == len(data)
    # ... and by the device's own inflater (msx_inflate.hip)
    if blocks:
        comp, tab, total = m.bgzf_blocks(payloads, datas)
        out, st, refused = m.bgzf_inflate(ctx, comp, tab, len(payloads), total)
        assert refused == 0 and not st.any() and out[:total].tobytes() == data


def twin(tmp_path):
    """the host twin of the encoder (tests/c/deflate_twin.c over msamtools_amd/csrc/msx_deflate_model.h)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "deflate_twin")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(root, "tests", "c", "deflate_twin.c"), "-lz"])

    def run(data, level=6):
        src, dst = str(tmp_path / "twin_in"), str(tmp_path / "twin_out")
        with open(src, "wb") as fh:
            fh.write(data)
        subprocess.check_call([exe, "-L", str(level), src, dst], stdout=subprocess.DEVNULL)
        with open(dst, "rb") as fh:
            return fh.read()
    return run


SIZES = [0, 1, 15, 16, 17, 4095, 4096, 4097, PAYLOAD - 1, PAYLOAD, PAYLOAD + 1, 2 * PAYLOAD, 3 * PAYLOAD + 17, 1_000_003]


@pytest.mark.parametrize("level", [0, 6, 1, 9])
def test_blocks_decode_to_the_input(ctx, level):
    rng = np.random.default_rng(41)
    for n in SIZES:
        for kind in ("random", "bam", "zeros"):
            data = (rng.integers(0, 256, n, dtype=np.uint8).tobytes() if kind == "random" else
                    bam_like(rng, n) if kind == "bam" else bytes(n))
            stream, n_blk = M().bgzf_deflate(ctx, data, level)
            check_stream(ctx, data, stream, n_blk, level)


@pytest.mark.parametrize("level", [0, 6])
def test_large_input(ctx, level):
    rng = np.random.default_rng(7)
    data = bam_like(rng, 40_000_000, seq=True)
    stream, n_blk = M().bgzf_deflate(ctx, data, level)
    assert n_blk == (len(data) + PAYLOAD - 1) // PAYLOAD
    assert gzip.decompress(stream) == data
    if level == 0:
        assert len(stream) == len(data) + 31 * n_blk


@pytest.mark.parametrize("level", [0, 6])
def test_emit_as_finished_blocks(ctx, level):
    """filter's output records of an unpacked batch, gathered and framed in one call: the blocks hold exactly the
    record stream msx_unpack_emit returns"""
    import struct
    m = M()
    rng = np.random.default_rng(3)
    recs = []
    for k in range(30000):
        name = b"r%07d\0" % (k // 3)
        body = struct.pack("<iiBBHHHIiii", 0, 10 + k, len(name), 30, 4680, 1, 0, 0, -1, -1, 0) + name + struct.pack("<I", 100 << 4)
        body += b"NMC\1ASC\x60MDZ" + (b"%d" % int(rng.integers(1, 100))) + b"\0"
        recs.append(struct.pack("<I", len(body)) + body)
    stream = b"".join(recs)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    up = m.Unpack(ctx)
    up.seed()
    up.enqueue(stream, pool_mode=1, n_targets=5, last=True)
    res, view = up.finish()
    n = int(res.n_records)
    assert n == len(recs)
    emit = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int32)
    d = ctx.alloc(4 * emit.size)
    ctx.to_dev(d, emit)
    want = b"".join(stream[int(off[i]):int(off[i + 1])] for i in emit)
    assert up.emit(d, emit.size, len(stream)) == want
    blocks, n_blk = up.emit_bgzf(d, emit.size, len(stream), level)
    check_stream(ctx, want, blocks, n_blk, level)
    # nothing emitted: no bytes, no blocks
    blocks, n_blk = up.emit_bgzf(d, 0, len(stream), level)
    assert blocks == b"" and n_blk == 0
    ctx.free(d)
    up.close()


def fibonacci_bytes(rng):
    fib = [1, 1]
    while sum(fib) + fib[-1] + fib[-2] < PAYLOAD:
        fib.append(fib[-1] + fib[-2])
    data = np.concatenate([np.full(f, i, np.uint8) for i, f in enumerate(fib)])
    rng.shuffle(data)
    return data.tobytes() * 2


@pytest.mark.parametrize("level", [6, 9, 2])
def test_the_kernel_writes_the_twins_bytes(ctx, tmp_path, level):
    """the device encoder against its one-position-at-a-time restatement on the host: the same blocks, bit for bit --
    matches, lazy decisions, trees, headers (what the lanes do side by side is what the model does in order); at the three
    geometries the level dials (window and table sizes: msx_bgzf_deflate_launch, df_opts_for_level)"""
    rng = np.random.default_rng(99)
    run = twin(tmp_path)
    cases = {
        "bam-like": bam_like(rng, 6 * PAYLOAD + 1234),
        "bam-like with seq/qual": bam_like(rng, 4 * PAYLOAD + 77, seq=True),
        "random (stored)": rng.integers(0, 256, 2 * PAYLOAD + 5, dtype=np.uint8).tobytes(),
        "nibbles (literals only)": rng.integers(0, 16, PAYLOAD + 999, dtype=np.uint8).tobytes(),
        "zeros": bytes(3 * PAYLOAD),
        "period 4": (b"\x12\x23\x34\x45" * 40000)[:2 * PAYLOAD + 3],
        "period 300": (rng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 500)[:PAYLOAD + 1],
        "tiny": b"abcabcabcabcabcabc",
        "one byte": b"Q",
        "text": (b"the quick brown fox jumps over the lazy dog; " * 3000)[:PAYLOAD * 2 - 1],
        "skewed (long codes)": bytes(np.minimum(rng.geometric(0.35, 3 * PAYLOAD), 255).astype(np.uint8)),
        # codes deeper than their limit: Fibonacci literal frequencies (literal/length code past 15 bits), and the block of
        # filter's output whose code-length code goes past 7 (the repair counts Kraft units: msx_deflate_model.h)
        "fibonacci literals": fibonacci_bytes(rng),
        "code-length code past its limit": open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                             "deflate_block_clcode_overflow.bin"), "rb").read(),
    }
    for name, data in cases.items():
        got, n_blk = M().bgzf_deflate(ctx, data, level)
        want = run(data, level)
        assert len(got) == len(want), (name, len(got), len(want))
        if got != want:
            first = next(i for i in range(len(got)) if got[i] != want[i])
            raise AssertionError(f"{name}: first difference at byte {first} of {len(got)}")


def test_ratio_next_to_zlib(ctx):
    """compressed size within 15 % of zlib level 6 on name-grouped BAM records (the reference's writer is htslib at level 6:
    msam_filter.c:464-470); output bytes are not pinned, record identity is"""
    rng = np.random.default_rng(5)
    data = bam_like(rng, 60 * PAYLOAD)
    got, n_blk = M().bgzf_deflate(ctx, data, 6)
    z = 0
    for i in range(0, len(data), PAYLOAD):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z += len(co.compress(data[i:i + PAYLOAD]) + co.flush()) + 26
    assert len(got) <= 1.15 * z, (len(got), z)


def test_encoder_beside_the_next_batch(ctx):
    """msx_unpack_emit_bgzf_enqueue / _complete: batch k is deflated on a stream of its own while batch k + 1 is walked and
    gathered; two batches in flight, completed in order -- the same blocks as the one-call form gives, batch by batch"""
    import struct
    m = M()
    rng = np.random.default_rng(17)

    def records(n, tag):
        recs = []
        for k in range(n):
            name = b"%s%07d\0" % (tag, k // 3)
            body = struct.pack("<iiBBHHHIiii", 0, 10 + k, len(name), 30, 4680, 1, 0, 0, -1, -1, 0) + name + struct.pack("<I", 100 << 4)
            body += b"NMC\1ASC\x60MDZ" + (b"%d" % int(rng.integers(1, 100))) + b"\0"
            recs.append(struct.pack("<I", len(body)) + body)
        return recs
    batches = [records(30000, b"a"), records(5000, b"b"), records(41000, b"c"), records(1, b"d"), records(20000, b"e")]
    up = m.Unpack(ctx)
    up.seed()
    want, dptrs, pending, got = [], [], [], []
    for i, recs in enumerate(batches):
        stream = b"".join(recs)
        off = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
        up.enqueue(stream, pool_mode=0, n_targets=5, last=True)
        res, view = up.finish()
        n = int(res.n_records)
        assert n == len(recs)
        emit = np.sort(rng.choice(n, size=max(1, n // 2), replace=False)).astype(np.int32) if i != 3 else np.zeros(0, np.int32)
        d = ctx.alloc(4 * max(emit.size, 1))
        if emit.size:
            ctx.to_dev(d, emit)
        dptrs.append(d)
        want.append(b"".join(stream[int(off[j]):int(off[j + 1])] for j in emit))
        up.emit_bgzf_enqueue(d, emit.size, 6)
        pending.append(i)
        if len(pending) == 2:                       # two in flight: complete the older one
            got.append(up.emit_bgzf_complete())
            pending.pop(0)
    while pending:
        got.append(up.emit_bgzf_complete())
        pending.pop(0)
    for i, (blocks, n_blk) in enumerate(got):
        assert (gzip.decompress(blocks) if blocks else b"") == want[i], i
        one, _ = m.bgzf_deflate(ctx, want[i], 6)
        assert blocks == one, i                     # the same encoder, the same bytes
    for d in dptrs:
        ctx.free(d)
    up.close()
    # the context remembered the closed unpacker's encoder stream as the last user of the encoder's scratch; a
    # later call that has to GROW that scratch drained the remembered stream -- a destroyed handle (a spurious MSX_ERR_HIP).
    # msx_unpack_destroy forgets it now: a larger input on the same context, then a new unpacker's encoder, both go through
    big = bytes(rng.integers(0, 4, 6_000_000, dtype=np.uint8))
    one, n_blk = m.bgzf_deflate(ctx, big, 6)
    assert gzip.decompress(one) == big and n_blk == (len(big) + PAYLOAD - 1) // PAYLOAD
    up2 = m.Unpack(ctx)
    up2.seed()
    recs = records(90000, b"z")
    stream = b"".join(recs)
    up2.enqueue(stream, pool_mode=0, n_targets=5, last=True)
    res, view = up2.finish()
    emit = np.arange(int(res.n_records), dtype=np.int32)
    d = ctx.alloc(4 * emit.size)
    ctx.to_dev(d, emit)
    up2.emit_bgzf_enqueue(d, emit.size, 6)
    blocks, _ = up2.emit_bgzf_complete()
    assert gzip.decompress(blocks) == stream
    ctx.free(d)
    up2.close()
