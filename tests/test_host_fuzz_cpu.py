"""Damaged BAM input is refused with the reader's own fatal errors, never by reading outside a record:
records whose announced fields (read name, CIGAR, SEQ/QUAL, aux) do not fit their length, aux fields that do
not end inside the record, truncated streams.  The BGZF layer's CRC would hide such records from a byte-flip
fuzzer, so the payload is damaged first and then wrapped into valid blocks."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _block(data):
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


@pytest.mark.skipif(not os.path.exists(BIN), reason="host binary not built")
def test_damaged_records_are_fatal_errors_not_overreads(tmp_path):
    src = tmp_path / "in.bam"
    with open(src, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "800", "--refs", "50", "-b"], stdout=fh)
    raw = gzip.open(src).read()
    l_text = struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, 8 + l_text)[0]
    p = 12 + l_text
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    rng = random.Random(20261003)
    seen = set()
    for it in range(60):
        d = bytearray(raw)
        mode = it % 4
        if mode == 0:
            q = rng.randrange(p, len(d))
            d[q] ^= 1 << rng.randrange(8)
        elif mode == 1:
            d = d[: rng.randrange(p, len(d))]
        elif mode == 2:
            q = rng.randrange(p, len(d))
            d[q:q + 4] = struct.pack("<I", rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 35]))
        else:
            q = rng.randrange(p, len(d))
            del d[q:q + rng.randint(1, 40)]
        bam = tmp_path / "m.bam"
        with open(bam, "wb") as fh:
            for i in range(0, len(d), 60000):
                fh.write(_block(bytes(d[i:i + 60000])))
            fh.write(EOF_BLOCK)
        for cmd in ([DEV, "recode", str(bam)], [DEV, "pipetest", "1", "1", str(bam)]):
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120,
                               env=dict(os.environ, MSX_THREADS="4", MSX_BATCH_BYTES="300000"))
            err = r.stderr.decode(errors="replace").strip()
            assert r.returncode in (0, 1), (cmd, r.returncode, err[-300:])      # never a signal
            if r.returncode == 1:
                assert err.startswith("Fatal Error: "), err[-300:]
                seen.add(err.split("(")[0][:40])
    assert any("Corrupt BAM record" in s for s in seen) and any("Truncated BAM record" in s for s in seen), seen


def test_damaged_compressed_sam_text_ends_the_run(tmp_path):
    """gzip-compressed SAM text cut anywhere or with a byte flipped: the run ends -- a fatal error or, where the damage leaves a
    valid stream of valid lines, a result -- within seconds, never a hang between the decompressor thread and the reader and
    never a signal (the thread writes into a pipe the reader owns)."""
    import gzip as gz
    bam, sam = tmp_path / "a.bam", tmp_path / "a.sam"
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "1500", "--refs", "30", "-b"], stdout=fh)
    with open(sam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", str(bam)], stdout=fh)
    text = open(sam, "rb").read()
    members = gz.compress(text[:len(text) // 2], 6) + gz.compress(text[len(text) // 2:], 6)
    rng = random.Random(77)
    outcomes = set()
    for it in range(40):
        d = bytearray(members)
        if it % 2:
            d = d[:rng.randrange(20, len(d))]
        else:
            d[rng.randrange(12, len(d))] ^= 1 << rng.randrange(8)
        path = tmp_path / "m.sam.gz"
        path.write_bytes(bytes(d))
        for cmd in ([DEV, "digest", str(path)], [DEV, "pipetest", "1", "1", str(path)]):
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=20,
                               env=dict(os.environ, MSX_THREADS="4", MSX_SAM_CHUNK="20000"))
            assert r.returncode in (0, 1), (it, cmd[1], r.returncode, r.stderr[-300:])
            outcomes.add(r.returncode)
    assert 1 in outcomes
