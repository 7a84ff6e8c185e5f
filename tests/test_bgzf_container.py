"""The same records in unusual BGZF containers (-m gpu): blocks of one byte, of 65 536 bytes (the format's most), empty blocks
in the middle, no EOF marker at the end, stored (level 0) blocks, the header cut across blocks -- through `filter --besthit`
and `profile`, with the blocks inflated on the device and on the host: the outputs of the plain file (htslib's bgzf_read takes
all of these: SAMv1 4.1; an absent EOF marker is a warning there, not an error)."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))
EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def block(chunk, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(chunk) + co.flush()
    assert len(comp) + 26 <= 65536, (len(chunk), len(comp))
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


def frame(raw, sizes, level=6, eof=True, empties_every=0):
    out, p, k = bytearray(), 0, 0
    while p < len(raw):
        n = sizes[k % len(sizes)]
        out += block(raw[p:p + n], level)
        p += n
        k += 1
        if empties_every and k % empties_every == 0:
            out += block(b"", level)                  # an empty block: ISIZE 0 (what an EOF marker is, in the middle)
    if eof:
        out += EOF_BLOCK
    return bytes(out)


@pytest.fixture(scope="module")
def base(tmp_path_factory):
    d = tmp_path_factory.mktemp("bgzf")
    src = str(d / "in.bam")
    with open(src, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "30000", "--refs", "400", "-b", "--seq"], stdout=fh)
    raw = gzip.open(src, "rb").read()
    want = {}
    want["filter"] = subprocess.check_output([BIN, "filter", "-l", "80", "-p", "95", "-z", "80", "--besthit", src])
    subprocess.check_call([BIN, "profile", "--label", "S", "-o", str(d / "p.gz"), src], stderr=subprocess.DEVNULL)
    want["profile"] = [l for l in gzip.open(str(d / "p.gz"), "rt").read().split("\n") if not l.startswith("# Command")]
    return d, raw, want


CASES = {
    "one_byte_blocks_in_the_header": lambda raw: frame(raw[:300], [1]) [:-28] + frame(raw[300:], [0xff00]),
    "small_odd_blocks": lambda raw: frame(raw, [1, 7, 311, 4099, 13], level=1),
    "largest_blocks": lambda raw: frame(raw, [65536]),
    "stored_blocks": lambda raw: frame(raw, [0xff00 - 7], level=0),
    "empty_blocks_in_the_middle": lambda raw: frame(raw, [20000, 3], empties_every=3),
    "no_eof_marker": lambda raw: frame(raw, [0xff00], eof=False),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("env", [{}, {"MSX_HOST_INFLATE": "1"}, {"MSX_BATCH_BYTES": "700000", "MSX_COMP_BLOCKS": "37"}], ids=["device", "host", "small_batches"])
def test_unusual_bgzf_containers(base, case, env):
    d, raw, want = base
    if case == "small_odd_blocks":             # (a block per seven bytes: keep the file short -- cut at a record's end)
        p = 8 + struct.unpack_from("<i", raw, 4)[0]
        n_ref = struct.unpack_from("<i", raw, p)[0]
        p += 4
        for _ in range(n_ref):
            p += 8 + struct.unpack_from("<i", raw, p)[0]
        while p < len(raw) // 6:
            p += 4 + struct.unpack_from("<i", raw, p)[0]
        raw = raw[:p]
        plain = str(d / "cut_plain.bam")
        open(plain, "wb").write(frame(raw, [0xff00]))
        want = {"filter": subprocess.check_output([BIN, "filter", "-l", "80", "-p", "95", "-z", "80", "--besthit", plain])}
        subprocess.check_call([BIN, "profile", "--label", "S", "-o", str(d / "cut_p.gz"), plain], stderr=subprocess.DEVNULL)
        want["profile"] = [l for l in gzip.open(str(d / "cut_p.gz"), "rt").read().split("\n") if not l.startswith("# Command")]
    path = str(d / f"{case}.bam")
    open(path, "wb").write(CASES[case](raw))
    e = dict(os.environ, **env)
    r = subprocess.run([BIN, "filter", "-l", "80", "-p", "95", "-z", "80", "--besthit", path], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (case, r.stderr.decode()[-600:])
    assert r.stdout == want["filter"], case
    out = str(d / f"{case}.p.gz")
    r = subprocess.run([BIN, "profile", "--label", "S", "-o", out, path], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (case, r.stderr.decode()[-600:])
    got = [l for l in gzip.open(out, "rt").read().split("\n") if not l.startswith("# Command")]
    assert got == want["profile"], case
