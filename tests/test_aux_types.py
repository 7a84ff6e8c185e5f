"""Aux fields of every BAM type around the three the path reads (SAMv1 4.2.4: A c C s S i I f Z H B): what aligners really
write -- bwa's XA:Z / SA:Z / MC:Z, minimap2's tp:A / de:f / cm:i, base-modification arrays ML:B:C, hex strings -- in front of,
between and behind NM / MD / AS.  The walkers (host: msh_io.c / msh_pipeline.c; device: msx_unpack.hip) must step over each by
its own size to find the tags, and filter's output must carry every byte of them.
  * no GPU: SAM text -> BAM -> SAM text is the identity (integers come back as SAM's ':i', floats through %g);
  * -m gpu: `filter` on the text, on the BAM with the device-side walk and with the host-side walk: the oracle's records."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
import samio
from conftest import ROOT

BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))

EXTRA = ["tp:A:P", "XA:Z:chr2,+100,50M,1;chr3,-7,48M2S,2;", "MC:Z:50M", "de:f:0.0123", "cm:i:12", "s1:i:-7", "XS:i:300", "XB:i:70000",
         "XN:i:-70000", "XU:i:4000000000", "ML:B:C,1,2,255", "ZS:B:s,-1,2,-32768", "ZI:B:I,0,4294967295", "ZF:B:f,0.5,1.5,-2", "ZC:B:c,-128,127",
         "XH:H:1AE301", "RG:Z:grp", "ZE:Z:", "ZW:B:S,65535", "ZJ:B:i,-2147483648,5"]


def make_sam(path, n_groups=4000, seed=3):
    rng = np.random.default_rng(seed)
    lines = ["@HD\tVN:1.6\tSO:queryname", "@SQ\tSN:chr1\tLN:100000", "@SQ\tSN:chr2\tLN:100000", "@SQ\tSN:chr3\tLN:50000"]
    for g in range(n_groups):
        for h in range(int(rng.integers(1, 5))):
            nm = int(rng.integers(0, 6))
            core = [f"NM:i:{nm}", f"MD:Z:{25 - nm}" + "A1" * nm + f"{25 - nm}" if nm else "MD:Z:50", f"AS:i:{50 - 2 * nm - int(rng.integers(0, 3))}"]
            if rng.random() < 0.15:
                core.pop(int(rng.integers(0, 2)))                      # NM or MD missing (one of them is enough)
            k = int(rng.integers(0, 9))
            extra = [EXTRA[i] for i in rng.choice(len(EXTRA), size=k, replace=False)]
            tags = list(extra)
            for c in core:                                             # the three tags at random places among the others
                tags.insert(int(rng.integers(0, len(tags) + 1)), c)
            flag = 0 if h == 0 else 256
            lines.append(f"q{g:06d}\t{flag}\tchr{int(rng.integers(1, 4))}\t{int(rng.integers(1, 40000))}\t60\t50M\t*\t0\t0\t{'ACGT' * 12}AC\t{'I' * 50}\t" + "\t".join(tags))
    open(path, "w").write("\n".join(lines) + "\n")
    return lines


def test_sam_bam_sam_round_trip_of_every_aux_type(tmp_path):
    sam = str(tmp_path / "aux.sam")
    lines = make_sam(sam, n_groups=1500)
    bam = str(tmp_path / "aux.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", sam], stdout=fh)
    back = subprocess.check_output([DEV, "recode", "-h", bam]).decode().split("\n")[:-1]
    assert back == lines
    # the pipeline's parallel parser and the record-at-a-time reader agree on every byte
    a = subprocess.check_output([DEV, "digest", "--full", sam], env=dict(os.environ, MSX_THREADS="6", MSX_SAM_CHUNK="20000")).decode()
    b = subprocess.check_output([DEV, "digest", "--full", bam]).decode()
    assert a == b


@pytest.mark.gpu
@pytest.mark.parametrize("opts,cli", [(dict(l=30, p=92, z=80, besthit=True), ["-l", "30", "-p", "92", "-z", "80", "--besthit"]),
                                      (dict(p=96), ["-p", "96"]), (dict(l=30, rescore=True, uniqhit=True), ["-l", "30", "--rescore", "--uniqhit"])])
def test_filter_steps_over_every_aux_type(tmp_path, opts, cli):
    sam = str(tmp_path / "aux.sam")
    lines = make_sam(sam)
    body = [l for l in lines if not l.startswith("@")]
    _, rec = samio.read_sam(sam)
    want = orc.run_filter(rec, **opts)
    assert want["rc"] == 0 and 0 < len(want["emit"]) < len(body)
    bam = str(tmp_path / "aux.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", sam], stdout=fh)

    def expect(i):
        if not opts.get("rescore"):
            return body[i]
        f = body[i].split("\t")                                          # msam_filter.c:160-168: first AS dropped, AS:i appended
        k = next(j for j, t in enumerate(f) if j >= 11 and t.startswith("AS:"))
        return "\t".join(f[:k] + f[k + 1:] + [f"AS:i:{int(want['as_out'][i])}"])
    exp = [expect(i) for i in want["emit"]]
    for src, env in ((["-S", sam], {}), ([bam], {"MSX_BATCH_RECORDS": "1500"}), ([bam], {"MSX_HOST_UNPACK": "1", "MSX_BATCH_RECORDS": "1500"})):
        r = subprocess.run([BIN, "filter"] + cli + src, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        assert r.stdout.decode().split("\n")[:-1] == exp, (src, env)
    # BAM out: every byte of the selected records (no --rescore: those are rewritten)
    if not opts.get("rescore"):
        out = str(tmp_path / "f.bam")
        with open(out, "wb") as fh:
            subprocess.check_call([BIN, "filter"] + cli + ["-b", bam], stdout=fh, env=dict(os.environ, MSX_BATCH_RECORDS="1500"))
        idx = str(tmp_path / "emit.u32")
        np.asarray(want["emit"]).astype("<u4").tofile(idx)
        assert subprocess.check_output([DEV, "digest", "--full", out]) == subprocess.check_output([DEV, "digest", "--full", "--select", idx, bam])


def make_long_cigar_sam(path, seed=5):
    """ordinary short records around three whose CIGAR has 70 000 operations (an ultra-long alignment): BAM holds such a CIGAR as
    the placeholder <l_seq>S<ref_len>N with the real one in a CG:B:I tag behind the other fields (SAMv1 4.2.2); htslib's reader
    swaps it back, so the reference's statistics come from the real one (msam_helper.c:246-268 -> mBamVector.c:23-133)"""
    rng = np.random.default_rng(seed)
    lines = ["@HD\tVN:1.6\tSO:queryname", "@SQ\tSN:chr1\tLN:1000000", "@SQ\tSN:chr2\tLN:1000000"]

    def short(q, flag=0):
        nm = int(rng.integers(0, 4))
        lines.append(f"{q}\t{flag}\tchr{int(rng.integers(1, 3))}\t{int(rng.integers(1, 40000))}\t60\t50M\t*\t0\t0\t{'ACGT' * 12}AC\t{'I' * 50}\t"
                     f"NM:i:{nm}\tAS:i:{50 - 2 * nm}\tXS:i:3")
    for g in range(300):
        short(f"a{g:05d}")
    long_ok = "1M1=" * 35000                    # 70 000 operations, 70 000 aligned bases, no edits: passes -l 80 -p 95 -z 80
    long_bad = "1M1D" * 35000                   # half of its aligned length are deletions: fails -p 95 on the REAL CIGAR
    seq70, seq35 = "ACGT" * 17500, "ACGT" * 8750
    lines.append(f"long1\t0\tchr1\t100\t60\t{long_ok}\t*\t0\t0\t{seq70}\t*\tNM:i:0\tXA:Z:x\tAS:i:70000")
    lines.append(f"long1\t256\tchr2\t100\t60\t{long_bad}\t*\t0\t0\t{seq35}\t*\tNM:i:35000\tAS:i:69000\tZI:B:I,1,2")
    lines.append(f"long2\t0\tchr2\t5000\t60\t{long_ok}\t*\t0\t0\t{seq70}\t*\tAS:i:7\tNM:i:1400\tMD:Z:70000")
    for g in range(300):
        short(f"b{g:05d}")
    # AS / NM of a non-integer type: htslib's bam_aux2i gives 0 (and sets errno) -- msam_filter.c:155,223
    lines.append("odd1\t0\tchr1\t7\t60\t50M\t*\t0\t0\t*\t*\tNM:i:0\tAS:f:49.5")
    lines.append("odd1\t256\tchr2\t7\t60\t50M\t*\t0\t0\t*\t*\tNM:i:0\tAS:i:-3")
    lines.append("odd2\t0\tchr1\t9\t60\t50M\t*\t0\t0\t*\t*\tNM:Z:seven\tAS:i:50")
    lines.append("odd2\t256\tchr1\t9\t60\t50M\t*\t0\t0\t*\t*\tNM:i:9\tAS:A:x")
    open(path, "w").write("\n".join(lines) + "\n")
    return lines


def test_a_cigar_of_70000_operations_goes_through_bam_and_back(tmp_path):
    sam = str(tmp_path / "long.sam")
    lines = make_long_cigar_sam(sam)
    bam = str(tmp_path / "long.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", sam], stdout=fh)
    # in the BAM: two CIGAR operations and a CG tag of 70 000 elements, the last of the record's fields
    import gzip
    import struct
    raw = gzip.open(bam, "rb").read()
    at = raw.index(b"long1\0") - 32
    n_cig, l_seq = struct.unpack_from("<H", raw, at + 12)[0], struct.unpack_from("<i", raw, at + 16)[0]
    c0, c1 = struct.unpack_from("<II", raw, at + 32 + 6)
    assert (n_cig, l_seq, c0, c1) == (2, 70000, 70000 << 4 | 4, 70000 << 4 | 3)
    bs = struct.unpack_from("<i", raw, at - 4)[0]
    assert raw[at + bs - 4 * 70000 - 8:at + bs - 4 * 70000] == b"CGBI" + struct.pack("<I", 70000)
    # ... and back: the real CIGAR in its place, no CG tag
    back = subprocess.check_output([DEV, "recode", "-h", bam]).decode().split("\n")[:-1]
    assert back == lines
    # the test-suite's own BAM reader makes the same swap
    _, a = samio.read_sam(sam)
    _, b = samio.read_bam(bam)
    assert (a.cigar_off == b.cigar_off).all() and (a.cigar == b.cigar).all() and (a.nm == b.nm).all() and (a.as_ == b.as_).all()


@pytest.mark.gpu
@pytest.mark.parametrize("opts,cli", [(dict(l=80, p=95, z=80, besthit=True), ["-l", "80", "-p", "95", "-z", "80", "--besthit"]),
                                      (dict(l=60000), ["-l", "60000"]), (dict(p=90, rescore=True, besthit=True), ["-p", "90", "--rescore", "--besthit"])])
def test_filter_reads_the_real_cigar_out_of_the_cg_tag(tmp_path, opts, cli):
    """text, BAM with the device-side walk, BAM with the host-side walk: the oracle's records -- computed from the 70 000
    operations, not from the two of the placeholder (on which long1 / long2 would fail -l 80: no aligned base at all)"""
    sam = str(tmp_path / "long.sam")
    lines = make_long_cigar_sam(sam)
    body = [l for l in lines if not l.startswith("@")]
    _, rec = samio.read_sam(sam)
    want = orc.run_filter(rec, **opts)
    names = [body[i].split("\t")[0] for i in want["emit"]]
    assert want["rc"] == 0 and "long1" in names and "long2" in names
    bam = str(tmp_path / "long.bam")
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-b", sam], stdout=fh)

    def expect(i):
        if not opts.get("rescore"):
            return body[i]
        f = body[i].split("\t")
        k = next(j for j, t in enumerate(f) if j >= 11 and t.startswith("AS:"))
        return "\t".join(f[:k] + f[k + 1:] + [f"AS:i:{int(want['as_out'][i])}"])
    exp = [expect(i) for i in want["emit"]]
    for src, env in ((["-S", sam], {}), ([bam], {"MSX_BATCH_RECORDS": "200"}), ([bam], {"MSX_HOST_UNPACK": "1", "MSX_BATCH_RECORDS": "200"})):
        r = subprocess.run([BIN, "filter"] + cli + src, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        assert r.stdout.decode().split("\n")[:-1] == exp, (src, env)
    if not opts.get("rescore"):
        # BAM out: the records' bytes as they came (placeholder and CG tag included)
        out = str(tmp_path / "f.bam")
        with open(out, "wb") as fh:
            subprocess.check_call([BIN, "filter"] + cli + ["-b", bam], stdout=fh, env=dict(os.environ, MSX_BATCH_RECORDS="200"))
        idx = str(tmp_path / "emit.u32")
        np.asarray(want["emit"]).astype("<u4").tofile(idx)
        assert subprocess.check_output([DEV, "digest", "--full", out]) == subprocess.check_output([DEV, "digest", "--full", "--select", idx, bam])
