"""A producer that trickles -- the reference's documented workflow is an aligner writing SAM into the pipe
(`bwa-mem2 mem ... | msamtools filter -S -bu ... -`, README.md:133-134 of the reference) -- must see its reads come out as they
are completed: the reference writes a read's alignments when the next read's first record arrives (msam_filter.c:120-125,186).
The pipeline here works in batches; a batch ends at its byte limit OR when the input has had nothing to give for MSX_IDLE_MS
(default 50 ms).  Before round 6 only the byte limit existed (96 MB): a slow producer saw nothing until it had sent that much
or closed the pipe."""
import os
import select
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))

HEADER = "@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:r1\tLN:100000\n"


def rec(i, h=0):
    return f"q{i:07d}\t{0 if h == 0 else 256}\tr1\t{10 + h}\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tAS:i:{50 - h}\n"


def run_with_pauses(cmd, pieces, pauses, env=None):
    """feeds `pieces` (bytes) to cmd's stdin, sleeping pauses[k] after piece k; returns (list of (time, bytes) as they came out of
    stdout, times at which the pieces had been written, return code)"""
    p = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    os.set_blocking(p.stdout.fileno(), False)
    got, sent = [], []
    t0 = time.monotonic()

    def drain(until):
        while True:
            left = until - time.monotonic()
            r, _, _ = select.select([p.stdout], [], [], max(left, 0))
            if r:
                b = p.stdout.read()
                if b:
                    got.append((time.monotonic() - t0, b))
                elif b == b"":
                    return
            if left <= 0:
                return
    for piece, pause in zip(pieces, pauses):
        p.stdin.write(piece)
        p.stdin.flush()
        sent.append(time.monotonic() - t0)
        drain(time.monotonic() + pause)
    p.stdin.close()
    end = time.monotonic() + 60
    while p.poll() is None and time.monotonic() < end:
        drain(time.monotonic() + 0.2)
    drain(time.monotonic() + 0.2)
    err = p.stderr.read().decode()
    assert p.wait(timeout=30) == 0, err[-800:]
    return got, sent, err


def first_time_of(got, needle):
    buf = b""
    for t, b in got:
        buf += b
        if needle in buf:
            return t
    return None


def test_sam_text_from_a_producer_that_pauses_comes_out_read_by_read():
    # batch 0 holds the preflight's window (100 000 records, as the reference's own look-ahead does); then a burst of ten reads,
    # three seconds of silence, the rest
    n0 = 120_000
    first = (HEADER + "".join(rec(i) + rec(i, 1) for i in range(n0 // 2))).encode()
    burst = "".join(rec(i) + rec(i, 1) for i in range(900_000, 900_010)).encode()
    rest = "".join(rec(i) for i in range(950_000, 950_100)).encode()
    got, sent, err = run_with_pauses([BIN, "filter", "-S", "-p", "95", "--besthit", "-"], [first, burst, rest], [2.0, 3.0, 0.0])
    out = b"".join(b for _, b in got).decode().split("\n")[:-1]
    assert len(out) == n0 // 2 + 10 + 100                      # --besthit: the better of each read's two alignments
    t_burst_written = sent[1]
    # the ninth read of the burst is complete once the tenth has begun: it must be out long before the producer resumes
    t9 = first_time_of(got, b"q0900008\t")
    t10 = first_time_of(got, b"q0900009\t")
    assert t9 is not None and t9 - t_burst_written < 1.0, (t9, t_burst_written, err[-400:])
    # ... and the tenth, still open while the producer is silent, only after it has gone on (the reference holds it as well)
    assert t10 is not None and t10 >= sent[2] - 0.05, (t10, sent)


def test_bam_blocks_from_a_producer_that_pauses(tmp_path):
    sam = str(tmp_path / "x.sam")
    n0 = 150_000
    with open(sam, "w") as f:
        f.write(HEADER)
        for i in range(n0 // 2):
            f.write(rec(i) + rec(i, 1))
    bam = str(tmp_path / "x.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", sam], stdout=fh)
    raw = open(bam, "rb").read()
    # cut the file at BGZF block boundaries: all but the last three data blocks, then two blocks, then the rest (with the EOF block)
    offs, p = [], 0
    while p < len(raw):
        offs.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    assert len(offs) > 12
    a, b = offs[-5], offs[-3]
    got, sent, err = run_with_pauses([BIN, "filter", "-p", "95", "--besthit", "-"], [raw[:a], raw[a:b], raw[b:]], [2.0, 3.0, 0.0],
                                     env={"MSX_TIMING": "1"})
    out = b"".join(b for _, b in got).decode().split("\n")[:-1]
    assert len(out) == n0 // 2
    # what the second piece completed is out before the third piece is sent
    n_before = sum(b.count(b"\n") for t, b in got if t < sent[2] - 0.05)
    n_first = sum(b.count(b"\n") for t, b in got if t < sent[1] - 0.05)
    assert n_before > n_first, (n_first, n_before, sent, err[-400:])
    assert max(t for t, b in got if t < sent[2] - 0.05) - sent[1] < 1.0
