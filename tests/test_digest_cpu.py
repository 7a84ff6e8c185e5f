"""`msamtools-dev digest --full [--select idx.u32]` (msh_dev.c): the whole-record digest the scale tests and bench.py hold
filter's output to -- every byte of every selected record, in the selection's order -- against a file assembled in
Python from the same records (no GPU)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_cli_scale import bgzf_blocks

DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(DEV):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "msamtools_amd", "csrc", "host")])


def digest(*args):
    out = subprocess.check_output([DEV, "digest", *args], env=dict(os.environ, MSX_THREADS="4", MSX_BATCH_RECORDS="9000")).decode().split()
    return int(out[0].split("=")[1]), out[1].split("=")[1]


def records_of(raw):
    p = 8 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    head, recs = raw[:p], []
    while p < len(raw):
        l = 4 + struct.unpack_from("<i", raw, p)[0]
        recs.append(raw[p:p + l])
        p += l
    return head, recs


@pytest.mark.parametrize("seq", [False, True])
def test_selected_full_digest_equals_the_digest_of_the_selection_written_out(tmp_path, seq):
    src = str(tmp_path / "in.bam")
    with open(src, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "8000", "--refs", "300", "-b"] + (["--seq"] if seq else []), stdout=fh)
    head, recs = records_of(gzip.open(src, "rb").read())
    rng = np.random.default_rng(5)
    keep = np.flatnonzero(rng.random(len(recs)) < 0.4)
    # filter's order is not the input's inside a pool (READ1 winners before READ2 winners): swap some neighbours
    sel = keep.copy()
    for k in range(0, len(sel) - 1, 7):
        sel[k], sel[k + 1] = sel[k + 1], sel[k]
    idx = str(tmp_path / "sel.u32")
    sel.astype("<u4").tofile(idx)
    out = str(tmp_path / "out.bam")
    open(out, "wb").write(bgzf_blocks(head + b"".join(recs[i] for i in sel)))
    for mode in (["--full"], []):
        want = digest(*mode, out)
        assert want[0] == len(sel)
        assert digest(*mode, "--select", idx, src) == want
        sel2 = sel.copy()
        sel2[3], sel2[10] = sel2[10], sel2[3]              # another order: another digest
        sel2.astype("<u4").tofile(idx + "2")
        assert digest(*mode, "--select", idx + "2", src) != want
    # one byte of one record's tail (aux / QUAL) changed: the plain digest does not see it, the full one does
    bad = bytearray(head + b"".join(recs[i] for i in sel))
    bad[len(bad) - 1] ^= 1              # (the last aux value byte)
    open(out, "wb").write(bgzf_blocks(bytes(bad)))
    assert digest(out) == digest("--select", idx, src)
    assert digest("--full", out) != digest("--full", "--select", idx, src)


def test_sam_text_with_many_references_is_not_a_linear_probe_per_record(tmp_path):
    """SAM text names its references; htslib resolves them through a hash (sam_hdr_name2tid).  200 k @SQ lines and a
    million records whose reference changes from one to the next: the name index (msh_hdr_name2tid) -- the linear probe it
    replaced took 40 s here, the index half a second.  Same records as the BAM they were written from."""
    import time
    bam, sam = str(tmp_path / "m.bam"), str(tmp_path / "m.sam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "200000", "--refs", "200000", "-b"], stdout=fh)
    with open(sam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", bam], stdout=fh)
    t0 = time.time()
    got = subprocess.run([DEV, "digest", sam], stdout=subprocess.PIPE, timeout=20).stdout.decode()
    assert got == subprocess.check_output([DEV, "digest", bam]).decode()
    assert time.time() - t0 < 20
    # and through the decode stage's parallel parser
    out = subprocess.check_output([DEV, "pipetest", "1", "1", sam], env=dict(os.environ, MSX_THREADS="8")).decode().split("\n")
    assert out[0].split()[1:5] == out[1].split()[1:5]


def test_compressed_sam_text_is_read_like_sam_text(tmp_path):
    """htslib's sam_open detects the format: gzip- and bgzip-compressed SAM text are read like plain SAM (the reference opens
    with "r" / "rb" and leaves the rest to it, msam_helper.c:203-215).  A decompressor thread feeds the text readers: one gzip
    member, one member per BGZF block, several members back to back, from a file and from a pipe -- every byte of every record
    as from the plain text; a stream cut inside a member or damaged is fatal, not a shorter input."""
    import gzip as gz
    bam, sam = str(tmp_path / "a.bam"), str(tmp_path / "a.sam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "5000", "--refs", "60", "-b", "--seq"], stdout=fh)
    with open(sam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", bam], stdout=fh)
    text = open(sam, "rb").read()
    want = digest("--full", sam)
    forms = {"one_member.sam.gz": gz.compress(text, 6), "bgzip.sam.gz": bgzf_blocks(text),
             "three_members.sam.gz": gz.compress(text[:1000], 1) + gz.compress(text[1000:70001], 9) + gz.compress(text[70001:], 6),
             # BGZF members (inflated side by side on the reader's pool), then a gzip member of another kind (one zlib stream
             # takes over where the first such member begins), then BGZF members again (which that stream reads like any other)
             "mixed.sam.gz": bgzf_blocks(text[:200000])[:-28] + gz.compress(text[200000:300000], 6) + bgzf_blocks(text[300000:])}
    for name, data in forms.items():
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        assert digest("--full", path) == want, name
        out = subprocess.run(f"cat {path} | {DEV} digest --full /dev/stdin", shell=True, stdout=subprocess.PIPE).stdout.decode().split()
        assert (int(out[0].split("=")[1]), out[1].split("=")[1]) == want, name
        # the decode stage's parallel parser on the same stream
        o = subprocess.check_output([DEV, "pipetest", "1", "1", path], env=dict(os.environ, MSX_THREADS="6", MSX_SAM_CHUNK="50000")).decode().split("\n")
        assert o[0].split()[1:5] == o[1].split()[1:5], name
    # BAM is still BAM (a QNAME or header that begins with 'B' does not make text of it, nor the reverse)
    assert digest(bam)[0] == want[0]
    cut = str(tmp_path / "cut.sam.gz")
    open(cut, "wb").write(forms["one_member.sam.gz"][:-3000])
    r = subprocess.run([DEV, "digest", cut], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"gzip stream" in r.stderr
    bad = bytearray(forms["one_member.sam.gz"])
    bad[len(bad) // 2] ^= 0x55
    open(cut, "wb").write(bytes(bad))
    r = subprocess.run([DEV, "digest", cut], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    # the same for bgzip'd text: cut inside a block, a damaged block (the block-parallel path reports through the same door)
    open(cut, "wb").write(forms["bgzip.sam.gz"][:-3000])
    r = subprocess.run([DEV, "digest", cut], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"gzip stream" in r.stderr
    bad = bytearray(forms["bgzip.sam.gz"])
    bad[len(bad) // 2] ^= 0x55
    open(cut, "wb").write(bytes(bad))
    r = subprocess.run([DEV, "digest", cut], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"gzip stream" in r.stderr


def test_cram_is_named_not_misparsed(tmp_path):
    """htslib reads CRAM; this reader does not, and says so (a QNAME that begins "CR" is still a QNAME)."""
    cram = tmp_path / "x.cram"
    cram.write_bytes(b"CRAM\x03\x00" + bytes(200))
    r = subprocess.run([DEV, "digest", str(cram)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"CRAM input is not supported" in r.stderr
    sam = tmp_path / "cr.sam"
    sam.write_text("CRR1\t0\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")
    assert digest(str(sam))[0] == 1
    out = subprocess.run(f"cat {sam} | {DEV} digest /dev/stdin", shell=True, stdout=subprocess.PIPE).stdout.decode()
    assert out.startswith("records=1 ")


def test_missing_eof_marker_is_htslibs_warning(tmp_path):
    """A seekable BAM without the BGZF end-of-file block: htslib warns when it reads the header and reads on (bgzf_check_EOF in
    bam_hdr_read); so does this reader, with that line; a pipe is not looked at."""
    bam = tmp_path / "a.bam"
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "synth", "--groups", "500", "--refs", "20", "-b"], stdout=fh)
    whole = bam.read_bytes()
    cut = tmp_path / "noeof.bam"
    cut.write_bytes(whole[:-28])
    a = subprocess.run([DEV, "digest", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([DEV, "digest", str(cut)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout
    assert a.stderr == b"" and b.stderr == b"[W::bam_hdr_read] EOF marker is absent. The input is probably truncated\n"
    c = subprocess.run(f"cat {cut} | {DEV} digest /dev/stdin", shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert c.stdout == a.stdout and c.stderr == b""
