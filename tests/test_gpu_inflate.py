"""BGZF inflate on the device (msx_inflate.hip) against zlib: the reference reads through htslib's BGZF layer
(msam_helper.c:246-268); every block is one raw DEFLATE stream whose CRC-32 and length are in the trailer."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def M():
    """(the library is loaded when a test first needs it, not while the test files are being collected: another test
    file imports torch during collection, and torch's bundled copy of the ROCm runtime must not be the second one in)"""
    import msamtools_amd
    return msamtools_amd


@pytest.fixture(scope="module")
def ctx():
    c = M().Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["lanes", "serial"])
def kernel(request, monkeypatch):
    """every test twice: the lane-parallel kernel (the default, k_bgzf_inflate_wave: the 64 lanes of a wave decode one deflate
    block's symbols at once; what it hands back goes to the serial kernel) and the serial kernel alone (MSX_INFLATE_SERIAL=1: one
    symbol after the other, rounds 3-5)"""
    if request.param == "serial":
        monkeypatch.setenv("MSX_INFLATE_SERIAL", "1")
    else:
        monkeypatch.delenv("MSX_INFLATE_SERIAL", raising=False)
    return request.param


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_at=(), flush_mode=zlib.Z_FULL_FLUSH, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    out, last = b"", 0
    for f in flush_at:
        out += co.compress(data[last:f]) + co.flush(flush_mode)
        last = f
    return out + co.compress(data[last:]) + co.flush()


def bam_like(rng, n_bytes):
    """records that repeat their neighbours, the way name-sorted BAM does"""
    out = bytearray()
    k = 0
    while len(out) < n_bytes:
        name = b"sim%08d" % k
        for h in range(int(rng.integers(1, 9))):
            core = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
            out += (60 + len(name)).to_bytes(4, "little") + core + name + b"\0" + bytes([100 << 4 & 255, 6, 0, 0])
            out += b"NMC" + bytes([int(rng.integers(0, 4))]) + b"ASC" + bytes([int(rng.integers(90, 101))]) + b"MDZ100\0"
        k += 1
    return bytes(out[:n_bytes])


def cases():
    rng = np.random.default_rng(20251003)
    c = []
    c.append(("empty", b"", {}))
    c.append(("one byte", b"A", {}))
    c.append(("zeros", bytes(65280), {}))
    c.append(("ones level 1", b"\x01" * 65536, dict(level=1)))
    for per in (2, 3, 5, 7, 16, 31, 63, 64, 65, 127, 300):
        unit = rng.integers(0, 256, per, dtype=np.uint8).tobytes()
        c.append((f"period {per}", (unit * (65280 // per + 1))[:65280], {}))
    c.append(("random (stored or nearly)", rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(), {}))
    c.append(("random level 0", rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(), dict(level=0)))
    c.append(("random nibbles, huffman only", rng.integers(0, 16, 65280, dtype=np.uint8).tobytes(), dict(strategy=zlib.Z_HUFFMAN_ONLY)))
    c.append(("skewed bytes (long codes)", (rng.geometric(0.08, 65280) % 256).astype(np.uint8).tobytes(), {}))
    c.append(("very skewed bytes (15-bit codes)", np.minimum(rng.geometric(0.5, 65280) * 3 + rng.integers(0, 200, 65280) * (rng.random(65280) < 0.002), 255).astype(np.uint8).tobytes(), dict(strategy=zlib.Z_HUFFMAN_ONLY)))
    for lvl in (1, 4, 6, 9):
        c.append((f"bam-like level {lvl}", bam_like(rng, 65280), dict(level=lvl)))
    c.append(("bam-like fixed codes", bam_like(rng, 65280), dict(strategy=zlib.Z_FIXED)))
    c.append(("bam-like rle", bam_like(rng, 65280), dict(strategy=zlib.Z_RLE)))
    c.append(("short fixed", b"hello hello hello hello", dict(strategy=zlib.Z_FIXED)))
    far = rng.integers(0, 256, 20000, dtype=np.uint8).tobytes()
    c.append(("far matches (20 KB back)", far + far + far[:25280], dict(level=9)))
    c.append(("far matches (32 KB back)", far[:16000] + bytes(16768) + far[:16000] + bytes(16000), dict(level=9)))
    mix = bam_like(rng, 30000) + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes() + bam_like(rng, 30280)
    c.append(("several blocks (full flush)", mix, dict(flush_at=(10000, 30000, 35000))))
    c.append(("several blocks (sync flush)", mix, dict(flush_at=(1, 2, 40000), flush_mode=zlib.Z_SYNC_FLUSH)))
    c.append(("small hash (memLevel 1: many blocks)", bam_like(rng, 65280), dict(mem=1)))
    c.append(("stored pieces", rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(), dict(level=0, flush_at=(100, 101, 30000))))
    c.append(("text", (b"the quick brown fox jumps over the lazy dog. " * 2000)[:65280], {}))
    c.append(("max block", bam_like(rng, 65536), {}))
    return c


def test_every_case_inflates_to_what_zlib_made_it_from(ctx):
    cs = cases()
    datas = [d for _, d, _ in cs]
    payloads = [raw_deflate(d, **kw) for _, d, kw in cs]
    for pl, d in zip(payloads, datas):
        assert zlib.decompress(pl, -15) == d
    for gap in (0, 1, 2, 3, 7):            # every alignment of the streams
        comp, blocks, total = M().bgzf_blocks(payloads, datas, gap=gap)
        out, st, refused = M().bgzf_inflate(ctx, comp, blocks, len(cs), total)
        bad = [(cs[i][0], int(st[i])) for i in range(len(cs)) if st[i] != 0]
        assert not bad and refused == 0, bad
        o = 0
        for (name, d, _) in cs:
            got = out[o:o + len(d)].tobytes()
            if got != d:
                first = next(k for k in range(len(d)) if got[k] != d[k])
                raise AssertionError(f"{name} (gap {gap}): first difference at byte {first} of {len(d)}")
            o += len(d)
        assert not out[total:].any()           # nothing written behind the last block


def test_many_blocks_of_a_bam_like_stream(ctx):
    rng = np.random.default_rng(7)
    stream = bam_like(rng, 3_000_000)
    datas = [stream[i:i + 65280] for i in range(0, len(stream), 65280)]
    payloads = [raw_deflate(d, level=(1, 6, 9)[i % 3]) for i, d in enumerate(datas)]
    comp, blocks, total = M().bgzf_blocks(payloads, datas)
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, len(datas), total)
    assert refused == 0 and not st.any()
    assert out[:total].tobytes() == stream


def test_damaged_blocks_are_refused_and_nothing_hangs(ctx):
    rng = np.random.default_rng(99)
    datas = [bam_like(rng, 40000) for _ in range(64)]
    good = [raw_deflate(d) for d in datas]
    payloads = []
    for i, pl in enumerate(good):
        b = bytearray(pl)
        if i % 4 == 1:
            for k in rng.integers(0, len(b), 3):
                b[int(k)] ^= 1 << int(rng.integers(0, 8))
        elif i % 4 == 2:
            b = b[:len(b) // 2]                                   # cut short
        elif i % 4 == 3:
            b = bytearray(rng.integers(0, 256, len(b), dtype=np.uint8).tobytes())   # noise
        payloads.append(bytes(b))
    comp, blocks, total = M().bgzf_blocks(payloads, datas)
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, len(datas), total)
    o = 0
    for i, d in enumerate(datas):
        if i % 4 == 0:
            assert st[i] == 0 and out[o:o + len(d)].tobytes() == d
        else:
            # a damaged stream either fails to decode or fails its CRC (zlib agrees that it is not the original)
            assert st[i] != 0, i
        o += len(d)
    assert refused == int((st != 0).sum())
    # a wrong CRC in the trailer alone
    comp, blocks, total = M().bgzf_blocks(good[:4], datas[:4], crcs=[zlib.crc32(d) ^ (i == 2) for i, d in enumerate(datas[:4])])
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, 4, total)
    assert list(st) == [0, 0, 8, 0] and refused == 1
    # a wrong length in the trailer
    comp, blocks, total = M().bgzf_blocks(good[:2], [datas[0], datas[1][:-5]], crcs=[zlib.crc32(datas[0]), zlib.crc32(datas[1])])
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, 2, total)
    assert st[0] == 0 and st[1] != 0


def test_random_streams_of_random_lengths(ctx):
    """400 blocks: length 1 .. 65536, data from five generators, every zlib level, strategy and memLevel at random"""
    rng = np.random.default_rng(424242)
    datas, payloads = [], []
    for k in range(400):
        n = int(rng.integers(1, 65537)) if k % 7 else int(rng.integers(1, 300))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            d = bam_like(rng, n)
        elif kind == 1:
            d = rng.integers(0, int(rng.integers(2, 257)), n, dtype=np.uint8).tobytes()
        elif kind == 2:
            unit = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes()
            d = (unit * (n // len(unit) + 1))[:n]
        elif kind == 3:
            base = bam_like(rng, max(n // 3, 1))
            d = (base + rng.integers(0, 256, max(n // 3, 1), dtype=np.uint8).tobytes() + base)[:n]
        else:
            d = np.minimum(rng.geometric(float(rng.uniform(0.02, 0.9)), n), 255).astype(np.uint8).tobytes()
        d = d.ljust(n, b"\0")[:n]
        flush_at = tuple(sorted(int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 3))))) if rng.random() < 0.3 else ()
        pl = raw_deflate(d, level=int(rng.integers(0, 10)),
                         strategy=[zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))],
                         flush_at=flush_at, flush_mode=[zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH][int(rng.integers(0, 2))],
                         mem=int(rng.integers(1, 10)))
        assert zlib.decompress(pl, -15) == d
        datas.append(d)
        payloads.append(pl)
    comp, blocks, total = M().bgzf_blocks(payloads, datas, gap=int(rng.integers(0, 5)))
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, len(datas), total)
    assert refused == 0 and not st.any(), [(i, int(v)) for i, v in enumerate(st) if v][:5]
    assert out[:total].tobytes() == b"".join(datas)


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nbits):                 # LSB first (header fields, extra bits)
        self.acc |= v << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, nbits):                # a Huffman code: most significant bit first
        self.put(int(format(c, f"0{nbits}b")[::-1], 2), nbits)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def canonical(lens):
    """code of every symbol with a length (RFC 1951, 3.2.2)"""
    codes, code = {}, 0
    for l in range(1, 16):
        for s, sl in enumerate(lens):
            if sl == l:
                codes[s] = (code, l)
                code += 1
        code <<= 1
    return codes


def dynamic_block(symbols, ll_lens, d_lens, last=True):
    """one dynamic-Huffman block from explicit code lengths (each length sent as itself: no run-length codes);
    symbols: ints (literal / end of block) or (length symbol, extra bits value, distance symbol, extra bits value)"""
    w = BitWriter()
    w.put(1 if last else 0, 1)
    w.put(2, 2)
    hlit, hdist = max(257, len(ll_lens)), max(1, len(d_lens))
    ll = list(ll_lens) + [0] * (hlit - len(ll_lens))
    dl = list(d_lens) + [0] * (hdist - len(d_lens))
    w.put(hlit - 257, 5)
    w.put(hdist - 1, 5)
    # the code-length code: every length 0..15 gets a 4-bit code (16 symbols of length 4: complete), 16-18 unused
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = [4] * 16 + [0, 0, 0]
    w.put(19 - 4, 4)
    for o in order:
        w.put(cl[o], 3)
    clc = canonical(cl)
    for l in ll + dl:
        w.code(*clc[l])
    llc, dc = canonical(ll), canonical(dl)
    lext = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
    dext = [0, 0, 0, 0] + [b for b in range(1, 14) for _ in (0, 1)]
    for sym in symbols:
        if isinstance(sym, tuple):
            ls, lx, ds, dx = sym
            w.code(*llc[ls])
            w.put(lx, lext[ls - 257])
            w.code(*dc[ds])
            w.put(dx, dext[ds])
        else:
            w.code(*llc[sym])
    return w.done()


def test_code_sets_zlib_never_writes(ctx):
    """Streams other encoders produce (htslib is often built with libdeflate): a block without any distance code, a
    distance code of a single symbol (incomplete: RFC 1951 allows exactly this), 15-bit literal codes next to 1-bit ones.
    zlib inflates them; so must the device -- or refuse, never mis-decode."""
    cases = []
    # 1. literals only, no distance code at all (HDIST = 1, its length 0)
    ll = [0] * 257
    for c in b"ACGT":
        ll[c] = 3
    ll[ord("N")] = 3
    ll[256] = 3
    ll[ord("\n")] = 3
    ll[ord("x")] = 3                      # eight codes of length 3: complete
    text = (b"ACGTNx\n" * 700)[:4321]
    cases.append((dynamic_block(list(text) + [256], ll, [0]), text))
    # 2. one distance symbol of length 1 (distance 1): runs
    ll = [0] * 286
    ll[ord("a")] = 2
    ll[ord("b")] = 2
    ll[256] = 2
    ll[285] = 2                           # length 258
    syms = [ord("a"), (285, 0, 0, 0), ord("b"), (285, 0, 0, 0), (285, 0, 0, 0), 256]
    cases.append((dynamic_block(syms, ll, [1]), b"a" * 259 + b"b" * 517))
    # 3. a very skewed literal code: lengths 1, 2, ..., 14, 15, 15
    ll = [0] * 257
    order = list(b"etaoinshrdlucmw") + [256]
    for k, c in enumerate(order):
        ll[c] = min(k + 1, 15)
    rng = np.random.default_rng(3)
    text = bytes(order[min(int(g) - 1, 14)] for g in rng.geometric(0.5, 20000))
    cases.append((dynamic_block(list(text) + [256], ll, [0]), text))
    # 4. 21 000 matches of two bits each (length 3, distance 1): one segment of the lane-parallel kernel holds more of them
    #    than its list of pieces does -- that kernel hands the block to the serial one
    ll = [0] * 258
    ll[257] = 1
    ll[ord("a")] = 2
    ll[256] = 2
    cases.append((dynamic_block([ord("a")] + [(257, 0, 0, 0)] * 21000 + [256], ll, [1]), b"a" * 63001))
    datas = [d for _, d in cases]
    payloads = [pl for pl, _ in cases]
    for pl, d in cases:
        assert zlib.decompress(pl, -15) == d
    comp, blocks, total = M().bgzf_blocks(payloads, datas)
    out, st, refused = M().bgzf_inflate(ctx, comp, blocks, len(datas), total)
    o = 0
    for i, d in enumerate(datas):
        if st[i] == 0:
            assert out[o:o + len(d)].tobytes() == d, i
        o += len(d)
    assert refused == int((st != 0).sum())
    assert not st.any(), list(st)          # all four are within what the device decodes itself


def test_blocks_written_by_libdeflate():
    """Streams from another encoder: libdeflate (what htslib and samtools deflate with in most builds) at levels 1, 6 and
    12 -- block splits, code lengths and match shapes zlib never chooses.  The compressed bytes are committed data
    (tests/golden/libdeflate_blocks.bin, made by tests/golden/make_libdeflate_blocks.py in the build container); the
    inputs are regenerated from that script's seed.  Every stream must inflate on the device to its input, CRC included."""
    import importlib.util
    import json
    import os
    m = M()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_libdeflate_blocks", os.path.join(here, "make_libdeflate_blocks.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    inputs = gen.inputs()
    blob = open(os.path.join(here, "libdeflate_blocks.bin"), "rb").read()
    index = json.load(open(os.path.join(here, "libdeflate_blocks.json")))["blocks"]
    assert len(index) == 21 and {b["level"] for b in index} == {1, 6, 12}
    payloads = [blob[b["offset"]:b["offset"] + b["length"]] for b in index]
    datas = [inputs[b["name"]] for b in index]
    for b, pl, d in zip(index, payloads, datas):
        assert len(d) == b["isize"] and zlib.crc32(d) == b["crc32"] and zlib.decompress(pl, -15) == d      # the fixture itself
    c = m.Context(0)
    try:
        comp, tab, total = m.bgzf_blocks(payloads, datas)
        out, st, refused = m.bgzf_inflate(c, comp, tab, len(payloads), total)
        bad = [(index[i]["name"], index[i]["level"], int(st[i])) for i in range(len(index)) if st[i]]
        assert refused == 0 and not bad, bad
        assert out[:total].tobytes() == b"".join(datas)
    finally:
        c.close()
