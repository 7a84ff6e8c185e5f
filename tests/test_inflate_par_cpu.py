"""The lane-parallel inflater's algorithm (msx_inflate.hip: k_bgzf_inflate_par) as restated on the host, one lane after the
other (msx_inflate_par_model.h): self-synchronising lanes, rounds, restarts with longer lanes, the end of a deflate block, the
windowed resolve with exact dependencies -- equal to zlib's inflate on valid streams of every level and strategy, and it
survives damaged ones (AddressSanitizer + UBSan build).  The kernel itself is checked on the device (tests/test_gpu_inflate.py)."""
import os
import subprocess
import zlib

import numpy as np

from conftest import ROOT


def _build(tmp_path, san=True):
    exe = str(tmp_path / "inflate_par_twin")
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if san else ["-O2"]
    subprocess.check_call(["gcc", *flags, "-std=gnu99", "-o", exe, os.path.join(ROOT, "tests", "c", "inflate_par_twin.c"), "-lz"])
    return exe


def test_the_lane_model_equals_zlib_and_survives_damage(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", IP_TWIN_CASES="250"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "rejected=0 bad=0" in out, out[-2000:]


def test_the_lane_model_on_bgzf_blocks_of_records_and_of_header_text(tmp_path):
    """name-grouped records (short tokens: the lanes fall onto the true chain within their 256 bits) and @SQ header text
    (matches of 258 bytes, 25 bits a token: too few tokens per lane, the segment is started again with longer lanes)"""
    exe = _build(tmp_path, san=False)
    rng = np.random.default_rng(11)
    recs = bytearray()
    k = 0
    while len(recs) < 600_000:
        name = b"sim%08d" % k
        for h in range(int(rng.integers(1, 9))):
            recs += (60 + len(name)).to_bytes(4, "little") + rng.integers(0, 256, 12, dtype=np.uint8).tobytes() + name + b"\0"
            recs += b"NMC" + bytes([int(rng.integers(0, 4))]) + b"ASC" + bytes([int(rng.integers(90, 101))]) + b"MDZ100\0"
        k += 1
    text = b"".join(b"@SQ\tSN:ref%07d\tLN:%d\n" % (i, 4496) for i in range(30000))
    path = str(tmp_path / "x.bgzf")
    with open(path, "wb") as f:
        for data in (bytes(recs), text):
            for o in range(0, len(data), 0xff00):
                d = data[o:o + 0xff00]
                co = zlib.compressobj(6, zlib.DEFLATED, -15)
                pl = co.compress(d) + co.flush()
                f.write(b"\x1f\x8b\x08\x04" + bytes(6) + b"\x06\x00BC\x02\x00" + (len(pl) + 25).to_bytes(2, "little") + pl
                        + zlib.crc32(d).to_bytes(4, "little") + len(d).to_bytes(4, "little"))
    r = subprocess.run([exe, path, "1000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "bad=0" in out, out[-2000:]
    assert "restarts with longer lanes" in out and ", 0 restarts" not in out, out[-600:]       # the header text needed them
