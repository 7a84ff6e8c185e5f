"""The host twin of the device DEFLATE encoder (tests/c/deflate_twin.c, msamtools_amd/csrc/msx_deflate_model.h): its
blocks are valid BGZF (the program inflates every block with zlib and compares), an independent reader (Python's gzip)
agrees, and the size stays within 15 % of zlib level 6 on name-grouped BAM-like records -- the writer the reference uses
is htslib at its default level (msam_filter.c:464-470).  The GPU test-suite asks the kernel for exactly these bytes."""
import gzip
import os
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAYLOAD = 0xff00


def bam_like(rng, n_bytes):
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


def test_twin_blocks_decode_and_stay_near_zlib(tmp_path):
    exe = str(tmp_path / "deflate_twin")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "c", "deflate_twin.c"), "-lz"])
    rng = np.random.default_rng(11)
    cases = {
        "bam": bam_like(rng, 12 * PAYLOAD + 321),
        "random": rng.integers(0, 256, PAYLOAD + 17, dtype=np.uint8).tobytes(),
        "zeros": bytes(2 * PAYLOAD),
        "skewed": bytes(np.minimum(rng.geometric(0.35, 2 * PAYLOAD), 255).astype(np.uint8)),
        "tiny": b"abcabcabcabc",
    }
    for name, data in cases.items():
        src, dst = str(tmp_path / "in"), str(tmp_path / "out")
        with open(src, "wb") as fh:
            fh.write(data)
        out = subprocess.check_output([exe, src, dst]).decode().split()
        assert int(out[0]) == len(data)
        with open(dst, "rb") as fh:
            stream = fh.read()
        assert len(stream) == int(out[1])
        assert gzip.decompress(stream) == data, name
        if name == "bam":
            z = 0
            for i in range(0, len(data), PAYLOAD):
                co = zlib.compressobj(6, zlib.DEFLATED, -15)
                z += len(co.compress(data[i:i + PAYLOAD]) + co.flush()) + 26
            assert len(stream) <= 1.15 * z, (len(stream), z)
        if name == "random":
            assert int(out[3]) >= 1                     # the full block is stored (the 17-byte tail is shorter with fixed codes)


def test_length_limited_codes_are_complete(tmp_path):
    """the tree builder's repair of codes deeper than their limit (15 bits; 7 for the code-length code): complete prefix
    codes under Fibonacci, power-of-two and one-giant frequency vectors (tests/c/huff_lengths_test.c), and the block of
    filter's output on which the first version wrote an over-subscribed code-length code (bench.py's digest check found
    it; committed as data: tests/golden/deflate_block_clcode_overflow.bin)"""
    exe = str(tmp_path / "huff_lengths_test")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "c", "huff_lengths_test.c")])
    assert subprocess.check_output([exe]).decode().strip() == "ok"
    twin = str(tmp_path / "deflate_twin")
    subprocess.check_call(["gcc", "-O2", "-o", twin, os.path.join(ROOT, "tests", "c", "deflate_twin.c"), "-lz"])
    blk = os.path.join(ROOT, "tests", "golden", "deflate_block_clcode_overflow.bin")
    out = str(tmp_path / "o")
    subprocess.check_call([twin, blk, out], stdout=subprocess.DEVNULL)            # (the twin inflates what it wrote with zlib)
    assert gzip.decompress(open(out, "rb").read()) == open(blk, "rb").read()


def test_levels_dial_the_geometry(tmp_path):
    """What the level means on the device is the encoder's window and table sizes (msx_bgzf_deflate_launch; the twin's
    df_opts_for_level): levels 1-3 the smallest, 4-6 what -b asks for (htslib's default 6, msam_filter.c:464-470), 7-9 round 4's
    encoder.  Every level's blocks decode (the twin inflates each block with zlib itself), a higher group writes no more than a
    fraction of a percent above a lower one, and every group stays within 15 % of zlib level 6 on name-grouped BAM-like records."""
    exe = str(tmp_path / "deflate_twin")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "c", "deflate_twin.c"), "-lz"])
    rng = np.random.default_rng(23)
    data = bam_like(rng, 20 * PAYLOAD + 99)
    src = str(tmp_path / "in")
    with open(src, "wb") as fh:
        fh.write(data)
    z = 0
    for i in range(0, len(data), PAYLOAD):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z += len(co.compress(data[i:i + PAYLOAD]) + co.flush()) + 26
    size = {}
    for level in (1, 3, 4, 6, 7, 9):
        dst = str(tmp_path / f"out{level}")
        out = subprocess.check_output([exe, "-L", str(level), src, dst]).decode().split()
        size[level] = int(out[1])
        with open(dst, "rb") as fh:
            assert gzip.decompress(fh.read()) == data, level
        assert size[level] <= 1.15 * z, (level, size[level], z)
    assert size[1] == size[3] and size[4] == size[6] and size[7] == size[9]          # three geometries, not nine
    # larger tables and a longer window find more (greedy matching is not monotone to the byte: on records this short the
    # three land within a percent of each other; on real BAM streams the gaps are 0.6 % and 0.7 %)
    assert size[9] <= 1.002 * size[6] and size[6] <= 1.002 * size[3]
    assert size[6] != size[3] and size[9] != size[6]                                   # ... and they are three encoders
