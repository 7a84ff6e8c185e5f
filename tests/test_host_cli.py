"""The C host command line (msamtools_amd/bin/msamtools).

CPU part: readers/writers (SAM text <-> BAM/BGZF round trips against the
independent Python parser), the reference's CLI validation messages and their
stdout/stderr split (tests/test_errors.sh), the QNAME preflight
(tests/test_qname_order.sh).  GPU part (-m gpu): the reference's Tier-1 filter /
besthit / profile / integration expectations end to end through the binary.
"""
import gzip
import json
import os
import subprocess
import time

import numpy as np
import pytest

import samio
from conftest import GOLDEN, ROOT, fixture_path

BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)
EXP = json.load(open(os.path.join(GOLDEN, "reference_expectations.json")))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "msamtools_amd", "csrc", "host")])


def run(args, stdin=None, env=None):
    e = dict(os.environ)
    e.update(env or {})
    exe = DEV if args and args[0] in ("synth", "digest", "recode", "pipetest", "restream", "rawtest", "keyorder") else BIN
    return subprocess.run([exe] + args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)


def body(path):
    return [l for l in open(path).read().split("\n") if l and not l.startswith("@")]


# ---- host I/O (no GPU) ------------------------------------------------------------

@pytest.mark.parametrize("fx", ["filter.sam", "cigar_eqx.sam", "besthit.sam", "long_qname.sam", "profile.sam",
                                "integration.sam", "coverage.sam"])
def test_sam_text_roundtrip(fx):
    r = run(["recode", fixture_path(fx)])
    assert r.returncode == 0 and r.stdout.decode().split("\n")[:-1] == body(fixture_path(fx))
    r = run(["recode", "-h", fixture_path(fx)])
    assert r.stdout.decode() == open(fixture_path(fx)).read()


@pytest.mark.parametrize("flag", ["-b", "-u"])
def test_bam_write_read_roundtrip(tmp_path, flag):
    src = fixture_path("besthit.sam")
    bam = tmp_path / "x.bam"
    bam.write_bytes(run(["recode", flag, src]).stdout)
    # an independent reader (python gzip + struct) sees the same records
    hdr, rec = samio.read_bam(str(bam))
    h2, want = samio.read_sam(src)
    assert hdr.target_name == h2.target_name and hdr.target_len == h2.target_len
    for k in ("flag", "tid", "pos", "cigar", "nm", "as_", "rflags"):
        assert (getattr(rec, k) == getattr(want, k)).all(), k
    assert bam.read_bytes().endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 0x1b, 0, 3, 0,
                                            0, 0, 0, 0, 0, 0, 0, 0]))          # BGZF EOF marker
    back = run(["recode", "-h", str(bam)])
    assert back.stdout.decode() == open(src).read()


def test_reads_reference_bam_and_stdin():
    path = fixture_path("tiny_aln.bam")
    out = run(["recode", path]).stdout.decode().split("\n")[:-1]
    hdr, rec = samio.read_bam(path)
    assert len(out) == rec.n == 16
    for i, line in enumerate(out):
        f = line.split("\t")
        assert f[0] == rec.name(i) and int(f[1]) == rec.flag[i] and f[5] == rec.cigar_str(i)
        assert f"MD:Z:{rec.md_str(i)}" in f and f"AS:i:{rec.as_[i]}" in f and f"NM:i:{rec.nm[i]}" in f
    piped = run(["recode", "-"], stdin=open(path, "rb").read())
    assert piped.stdout.decode().split("\n")[:-1] == out
    for threads in ("1", "4"):
        assert run(["recode", path], env={"MSX_THREADS": threads}).stdout.decode().split("\n")[:-1] == out


def test_multi_block_bam_roundtrip(tmp_path):
    """Enough records for many BGZF blocks and several inflate batches."""
    hdr = "@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:A\tLN:100000\n"
    lines = [f"r{i:07d}\t{0 if i % 3 else 256}\tA\t{1 + i % 9000}\t60\t50M\t*\t0\t0\t{'ACGT' * 12}AC\t{'I' * 50}\tNM:i:{i % 5}"
             f"\tMD:Z:50\tAS:i:{50 - i % 7}\tXX:Z:tag{i}\tXB:B:s,1,-2,{i % 100}" for i in range(60000)]
    sam = tmp_path / "big.sam"
    sam.write_text(hdr + "\n".join(lines) + "\n")
    bam = tmp_path / "big.bam"
    bam.write_bytes(run(["recode", "-b", str(sam)]).stdout)
    assert bam.stat().st_size < sam.stat().st_size / 3
    assert run(["recode", str(bam)]).stdout.decode().split("\n")[:-1] == lines
    assert len(gzip.open(str(bam)).read()) > 6_000_000


# ---- CLI validation (tests/test_errors.sh) ------------------------------------------

ERR_CASES = [
    (["filter", "-S", "-h", "FX"], "needs -l, -p, --ppt, -z, --besthit or --uniqhit", "stdout"),
    (["filter", "-S", "--besthit", "--uniqhit", "FX"], "--besthit cannot be combined with --uniqhit", "stdout"),
    (["filter", "-S", "-p", "95", "--ppt", "950", "FX"], "-p cannot be combined with --ppt", "stdout"),
    (["filter", "-S", "-p", "101", "FX"], "-p must be in the range [0,100]", "stdout"),
    (["filter", "-S", "-p", "-1", "FX"], "-p must be in the range [0,100]", "stdout"),
    (["filter", "-S", "-l", "-1", "FX"], "-l must be a non-negative integer", "stdout"),
    (["filter", "-S", "-z", "101", "FX"], "-z must be in the range [0,100]", "stdout"),
    (["filter", "-S", "-z", "-1", "FX"], "-z must be in the range [0,100]", "stdout"),
    (["filter", "-S", "-v", "--besthit", "FX"], "--invert cannot be combined with --besthit or --uniqhit", "stdout"),
    (["filter", "-S", "--ppt", "1001", "FX"], "--ppt must be in the range [-1000,1000]", "stdout"),
    (["profile", "-S", "FX"], "Use --help for usage instructions", "stdout"),
    (["profile", "-S", "--label", "x", "--pandas", "--no-pandas", "-o", "/dev/null", "FX"],
     "--pandas and --no-pandas cannot be used together", "stdout"),
    (["profile", "-S", "--label", "x", "--mincount", "-1", "-o", "/dev/null", "FX"],
     "--mincount must be a non-negative integer", "stdout"),
    (["profile", "-S", "--label", "x", "--total", "0", "-o", "/dev/null", "FX"],
     "--total must be a positive integer", "stdout"),
    (["definitely-not-a-command"], "unrecognized command", "stderr"),
]


@pytest.mark.parametrize("args,msg,stream", ERR_CASES, ids=[" ".join(c[0][:4]) for c in ERR_CASES])
def test_cli_validation_messages(args, msg, stream):
    r = run([fixture_path("filter.sam") if a == "FX" else a for a in args])
    assert r.returncode == 1
    assert msg in getattr(r, stream).decode()
    assert "@HD" not in r.stdout.decode()          # no header before validation (test_errors.sh:34-36)


def test_help_exits_zero():
    for sub in ("filter", "profile"):
        r = run([sub, "--help"])
        assert r.returncode == 0 and b"Usage:" in r.stdout
    assert run(["help"]).returncode == 0


def test_qname_preflight_rejects_bad_order(tmp_path):
    # tests/test_qname_order.sh:29-63
    r = run(["filter", "-S", "--besthit", fixture_path("qname_coordinate.sam")])
    assert r.returncode == 1 and b"declares 'SO:coordinate'" in r.stderr and r.stdout == b""
    r = run(["filter", "-S", "--besthit", fixture_path("qname_reopened.sam")])
    assert r.returncode == 1 and b"is not grouped by QNAME" in r.stderr and b"reappears at record 3" in r.stderr
    out = tmp_path / "p.gz"
    r = run(["profile", "-S", "--label", "x", "-o", str(out), fixture_path("qname_reopened.sam")])
    assert r.returncode == 1 and not out.exists()


def test_profile_refuses_a_leaked_world_size(tmp_path):
    """with WORLD_SIZE > 1 in the environment and a plain input path every rank would read the whole
    file and the all-reduce would multiply every count.  The command refuses (before touching input or GPU) unless
    the run is asked for as a rank ("{rank}" in the path / MSX_DIST=1) or explicitly as a single process (MSX_DIST=0)."""
    out = tmp_path / "p.gz"
    env = {"WORLD_SIZE": "4", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29777"}
    r = run(["profile", "-S", "--label", "x", "-o", str(out), fixture_path("profile.sam")], env=env)
    assert r.returncode == 1 and b'has no "{rank}"' in r.stderr and b"Fatal Error" in r.stderr and not out.exists()
    # WORLD_SIZE alone (no RANK: not a launcher's environment) is ignored, as is anything under MSX_DIST=0:
    # both get as far as needing the GPU (or succeed where there is one)
    for e in ({"WORLD_SIZE": "4"}, dict(env, MSX_DIST="0")):
        r = run(["profile", "-S", "--label", "x", "-o", str(out), fixture_path("profile.sam")], env=e)
        assert b'has no "{rank}"' not in r.stderr


# ---- end to end on the GPU ---------------------------------------------------------------

def cli_opts(opts):
    a = []
    for k, v in opts.items():
        if k in ("l", "p", "z"):
            a += [f"-{k}", str(v)]
        elif k == "ppt":
            a += ["--ppt", str(v)]
        elif v:
            a += [{"invert": "-v", "keep_unmapped": "-k"}.get(k, f"--{k}")]
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXP["filter"], ids=[c["name"] for c in EXP["filter"]])
def test_cli_filter_golden(case):
    r = run(["filter", "-S", "-h"] + cli_opts(case["opts"]) + [fixture_path(case["fixture"])])
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().split("\n")
    recs = [l.split("\t") for l in lines if l and not l.startswith("@")]
    assert ",".join(f"{f[0]}:{f[1]}" for f in recs) == case["records"], case["src"]
    pg = [l for l in lines if l.startswith("@PG")]
    assert len(pg) == 1 and "\tPN:msamtools\t" in pg[0] and "CL:msamtools filter -S -h" in pg[0]
    if case["opts"].get("besthit") or case["opts"].get("uniqhit"):
        assert "QNAME grouping check: confirmed by input header SO:queryname" in pg[0]     # test_besthit.sh:36-38
    else:
        assert "QNAME grouping check: not required for this operation" in pg[0]
    for key, val in case.get("as", {}).items():
        hit = [f for f in recs if f"{f[0]}:{f[1]}" == key]
        assert hit and all(f"AS:i:{val}" in f and sum(x.startswith("AS:") for x in f) == 1 for f in hit)


@pytest.mark.gpu
def test_cli_filter_bam_in_bam_out(tmp_path):
    t = EXP["tiny_aln"]
    src = fixture_path(t["fixture"])
    args = ["filter", "-b", "-l", "80", "-p", "95", "-z", "80", "--besthit", src]
    out = run(args, env={"MSX_BATCH_RECORDS": "3"})        # tiny batches: pools carried across batch cuts
    assert out.returncode == 0, out.stderr.decode()
    bam = tmp_path / "f.bam"
    bam.write_bytes(out.stdout)
    got = run(["recode", str(bam)]).stdout.decode().split("\n")[:-1]
    want = run(["recode", src]).stdout.decode().split("\n")[:-1]
    assert got == [want[i] for i in t["emit"]]             # records byte-identical as text, reference order
    ub = run(["filter", "-bu", "-l", "80", "-p", "95", "-z", "80", "--besthit", src])
    (tmp_path / "u.bam").write_bytes(ub.stdout)
    assert run(["recode", str(tmp_path / "u.bam")]).stdout.decode().split("\n")[:-1] == got


@pytest.mark.gpu
@pytest.mark.parametrize("cmd", ["filter", "filter_tee", "profile", "coverage", "summary"])
def test_cli_orderly_shutdown(tmp_path, synth_bams, cmd):
    """MSX_CLEAN_EXIT=1 (the leak-check / profiler mode: handles destroyed, input closed, main returns instead of _exit)
    ends with status 0 and the same output as the default quick exit, for every command."""
    args = {"filter": ["filter", "-bu", "-l", "80", "-p", "95", "-z", "80", "--besthit", synth_bams["b"]],
            "filter_tee": ["filter", "-bu", "-l", "80", "-p", "95", "-z", "80", "--besthit", "--profile-out", str(tmp_path / "p.gz"),
                           "--label", "S", synth_bams["b"]],
            "profile": ["profile", "--label", "S", "--multi", "prop", "-o", str(tmp_path / "o.gz"), synth_bams["b"]],
            "coverage": ["coverage", "--summary", "-o", str(tmp_path / "c.gz"), synth_bams["b"]],
            "summary": ["summary", synth_bams["b"]]}[cmd]
    a = run(args)
    b = run(args, env={"MSX_CLEAN_EXIT": "1", "MSX_TIMING": "1"})
    assert a.returncode == 0 and b.returncode == 0, (a.stderr.decode()[-400:], b.stderr.decode()[-800:])
    assert a.stdout == b.stdout


@pytest.mark.gpu
def test_cli_reads_compressed_sam_text(tmp_path, synth_bams):
    """`filter` and `profile` on gzip- and bgzip-compressed SAM text, named or on stdin, with -S and without (htslib detects
    the format either way: validate_profiles.py:735-751 feeds .sam without -S): what the plain text gives."""
    import gzip as gz
    sam = str(tmp_path / "in.sam")
    with open(sam, "wb") as fh:
        fh.write(run(["recode", "-h", synth_bams["b"]]).stdout)
    text = open(sam, "rb").read()
    gzp, bgz = str(tmp_path / "in.sam.gz"), str(tmp_path / "in.bgz.sam.gz")
    open(gzp, "wb").write(gz.compress(text, 5))
    from test_cli_scale import bgzf_blocks
    open(bgz, "wb").write(bgzf_blocks(text))
    filt = ["filter", "-l", "80", "-p", "95", "-z", "80", "--besthit"]
    want = run(filt + ["-S", sam])
    assert want.returncode == 0 and len(want.stdout) > 10000
    for path in (gzp, bgz):
        for s_flag in (["-S"], []):
            r = run(filt + s_flag + [path], env={"MSX_SAM_CHUNK": "300000"})
            assert r.returncode == 0 and r.stdout == want.stdout, (path, s_flag, r.stderr.decode()[-300:])
        r = run(filt + ["-S", "-"], stdin=open(path, "rb").read())
        assert r.returncode == 0 and r.stdout == want.stdout, path
    a, b = str(tmp_path / "a.gz"), str(tmp_path / "b.gz")
    assert run(["profile", "-S", "--label", "S", "-o", a, sam]).returncode == 0
    assert run(["profile", "--label", "S", "-o", b, "-"], stdin=open(gzp, "rb").read()).returncode == 0
    text_of = lambda p: [l for l in gzip.open(p, "rt").read().split("\n") if not l.startswith("# Command")]
    assert text_of(a) == text_of(b)


@pytest.mark.gpu
def test_cli_fatal_error_does_not_wait_for_a_silent_producer(tmp_path):
    """A fatal record found on the device while the reader sits in a read of a pipe whose other end has gone silent: the
    command dies with the reference's message at once (msam_filter.c:150-152).  mDie used to fflush(NULL) -- which locks every
    stream, the input's too, and that lock is held for as long as the read waits -- so the error waited for the producer."""
    import time
    text = tmp_path / "in.sam"                 # sent in one go (the silence must begin before the device has seen batch 0)
    text.write_text("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:r1\tLN:1000\n" +
                    "".join(f"q{i:07d}\t0\tr1\t10\t255\t10M\t*\t0\t0\t*\t*\tAS:i:5\n" for i in range(200000)))
    script = tmp_path / "slow.sh"
    script.write_text(f"#!/bin/bash\ncat {text}\nsleep 25\n")
    prod = subprocess.Popen(["bash", str(script)], stdout=subprocess.PIPE)
    t0 = time.time()
    cons = subprocess.Popen([BIN, "filter", "-S", "-p", "95", "-"], stdin=prod.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSX_SAM_CHUNK="1000000", MSX_BATCH_BYTES="2000000"))    # (a batch is cut by bytes: 96 MB by default)
    prod.stdout.close()
    try:
        _, err = cons.communicate(timeout=120)
        took = time.time() - t0
    finally:
        prod.kill()
        prod.wait()
    assert cons.returncode == 1 and b"Fatal Error" in err, err[-300:]
    assert took < 15, took                                   # (the producer would have gone on sleeping until 25 s)


def read_profile(path):
    text = gzip.open(path, "rt").read()
    head = [l for l in text.split("\n") if l.startswith("#")]
    rows = [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]
    return head, rows


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXP["profile"], ids=[c["name"] for c in EXP["profile"]])
def test_cli_profile_golden(case, tmp_path):
    src = fixture_path(case["fixture"])
    if "filter_opts" in case:      # test_integration.sh:49-69: filter output fed back to profile
        f = run(["filter", "-S", "-h"] + cli_opts(case["filter_opts"]) + [src])
        assert f.returncode == 0
        src = str(tmp_path / "filtered.sam")
        open(src, "wb").write(f.stdout)
    out = str(tmp_path / "p.tsv.gz")
    args = ["profile", "-S", "--label", "test", "--unit", case["unit"], "--multi", case["multi"], "--pandas", "-o", out]
    if case["nolen"]:
        args.append("--nolen")
    if case["total"] > 0:
        args += ["--total", str(case["total"])]
    if "mincount" in case:
        args += ["--mincount", str(case["mincount"])]
    r = run(args + [src])
    assert r.returncode == 0, r.stderr.decode()
    head, rows = read_profile(out)
    assert rows[0] == ["ID", "test"]
    vals = {r[0]: float(r[1]) for r in rows[1:]}
    for feat, (want, tol) in case["values"].items():
        assert abs(vals[feat] - want) <= tol, (feat, case["src"])
    text = "\n".join(head)
    assert "QNAME grouping check: confirmed by input header SO:queryname" in text
    if case["name"] in ("all", "equal", "ignore", "proportional"):     # test_profile.sh:39-46
        for s in ("Total inserts       : 7", "Mapped inserts      : 7", "- Multiple mapped : 1", "- Uniquely mapped : 6"):
            assert s in text
    if case["name"] in ("empty", "unmapped"):                           # test_profile.sh:119-126
        for s in ("Mapped inserts      :       0", "- Multiple mapped :       0", "- Uniquely mapped :       0",
                  "Effective inserts   :          0"):
            assert s in text


@pytest.mark.gpu
def test_cli_profile_header_modes_and_pipe(tmp_path):
    src = fixture_path("profile.sam")
    base = ["profile", "-S", "--unit", "ab", "--nolen", "--total", "7", "--multi", "equal"]
    out = str(tmp_path / "a.gz")
    assert run(base + ["--label", "default_output", "-o", out, src]).returncode == 0
    assert read_profile(out)[1][0] == ["ID", "default_output"]          # test_profile.sh:175-199
    assert run(base + ["--label", "legacy_output", "--no-pandas", "-o", out, src]).returncode == 0
    assert read_profile(out)[1][0] == ["legacy_output"]                 # :226-246
    # filter -bu ... | profile -   (README pipe; uncompressed BAM on stdin, gz profile on stdout)
    t = EXP["tiny_aln"]
    f = run(["filter", "-bu", "-l", "80", "-p", "95", "-z", "80", "--besthit", fixture_path(t["fixture"])])
    p = run(["profile", "--label", "S", "-o", "-", "-"], stdin=f.stdout)
    assert p.returncode == 0, p.stderr.decode()
    text = gzip.decompress(p.stdout).decode()
    rows = {l.split("\t")[0]: l.split("\t")[1] for l in text.split("\n") if l and not l.startswith("#")}
    for k, v in t["rel_values"].items():
        assert float(rows[k]) == pytest.approx(v, rel=1e-7)
    assert sum(1 for k, v in rows.items() if k not in ("ID",) and float(v) != 0) == 4
    assert "Purged inserts      :          3" in text and "Effective inserts   :          4" in text
    assert b"# Purged 3 inserts that mapped to features without unique inserts." in p.stderr
    assert b"PropSharing Iteration:  1; DELTA^2=0. CONVERGED!" in p.stderr


@pytest.mark.gpu
def test_cli_streaming_boundary(tmp_path):
    """tests/test_streaming.sh: a QNAME group straddling the 100 000-record preflight window."""
    lines = ["@SQ\tSN:A\tLN:1000", "@SQ\tSN:B\tLN:1000"]
    for i in range(1, 100000):
        lines.append(f"r{i:06d}\t0\tA\t100\t60\t10M\t*\t0\t0\tAAAAAAAAAA\tIIIIIIIIII\tAS:i:10\tNM:i:0")
    lines.append("boundary\t0\tA\t300\t60\t10M\t*\t0\t0\tAAAAAAAAAA\tIIIIIIIIII\tAS:i:5\tNM:i:0")
    lines.append("boundary\t256\tB\t400\t60\t10M\t*\t0\t0\tAAAAAAAAAA\tIIIIIIIIII\tAS:i:9\tNM:i:0")
    r = run(["filter", "-S", "--besthit", "-"], stdin=("\n".join(lines) + "\n").encode())
    assert r.returncode == 0, r.stderr.decode()[:500]
    out = r.stdout.decode().split("\n")[:-1]
    assert len(out) == 100000
    last = out[-1].split("\t")
    assert (last[0], last[1], last[2], last[3]) == ("boundary", "256", "B", "400")


def test_coverage_cli_validation():
    # tests/test_errors.sh (coverage block): -w must be positive, -o required
    r = run(["coverage", "-S", "-w", "0", "-o", "/dev/null", fixture_path("coverage.sam")])
    assert r.returncode == 1 and b"-w must be a non-zero positive integer" in r.stdout
    r = run(["coverage", "-S", fixture_path("coverage.sam")])
    assert r.returncode == 1
    assert run(["coverage", "--help"]).returncode == 0


@pytest.mark.gpu
def test_cli_coverage_golden(tmp_path):
    """tests/test_coverage.sh:31-80: per-position text with -w 4, --summary, --skipuncovered."""
    blk = EXP["coverage"]
    src = fixture_path(blk["fixture"])
    out = str(tmp_path / "positions.gz")
    r = run(["coverage", "-S", "-w", "4", "-o", out, src])
    assert r.returncode == 0 and r.stdout == b"" and r.stderr == b""
    want = []
    for name in ("A", "B", "C", "D"):
        v = blk["positions"][name]
        want.append(">" + name)
        for i in range(0, len(v), 4):
            want.append(" ".join(str(x) for x in v[i:i + 4]))
    assert gzip.open(out, "rt").read().split("\n")[:-1] == want
    r = run(["coverage", "-S", "--summary", "-o", out, src])
    assert r.returncode == 0
    assert gzip.open(out, "rt").read().split("\n")[:-1] == [f"{k}\t{v[0]}\t{v[1]}" for k, v in blk["summary"].items()]
    r = run(["coverage", "-S", "--summary", "--skipuncovered", "-o", out, src])
    assert gzip.open(out, "rt").read().split("\n")[:-1] == \
        [f"{k}\t{v[0]}\t{v[1]}" for k, v in blk["summary"].items() if k != "B"]
    r = run(["coverage", "-S", "-x", "-o", "-", src])
    text = gzip.decompress(r.stdout).decode()
    assert ">B" not in text and ">A" in text


@pytest.mark.gpu
@pytest.mark.parametrize("multi", ["proportional", "all"])
def test_cli_profile_genome_order_and_values(tmp_path, multi):
    """profile --genome (msam_profile.c:760-852): features = genomes in the key order of the reference's
    hash table (12 genomes: the 8th re-shuffles the first eight), values = the oracle's profile over
    the tid -> genome map."""
    import ctypes as C
    import oracle_lib as orc
    rnd = np.random.RandomState(5)
    n_seq, n_gen = 40, 12
    seq = [f"contig_{i:03d}" for i in range(n_seq)]
    gen_of = [f"strain_{(i * 7) % n_gen:02d}" for i in range(n_seq)]
    lens = [int(x) for x in rnd.randint(500, 3000, n_seq)]
    lines = ["@HD\tVN:1.6\tSO:queryname"] + [f"@SQ\tSN:{s}\tLN:{l}" for s, l in zip(seq, lens)]
    for r in range(400):
        hits = rnd.choice(n_seq, size=int(rnd.choice([1, 1, 1, 2, 3])), replace=False)
        for h in hits:
            lines.append(f"read{r:04d}\t0\t{seq[h]}\t{int(rnd.randint(1, 400))}\t60\t50M\t*\t0\t0\t" + "A" * 50 + "\t" + "I" * 50 +
                         "\tAS:i:50\tNM:i:0\tMD:Z:50")
    sam = str(tmp_path / "g.sam")
    open(sam, "w").write("\n".join(lines) + "\n")
    gdef = str(tmp_path / "genomes.tsv")
    order_in_file = list(rnd.permutation(n_seq))
    open(gdef, "w").write("".join(f"{gen_of[i]}\t{seq[i]}\n" for i in order_in_file))
    out = str(tmp_path / "p.gz")
    r = run(["profile", "-S", "--label", "t", "--unit", "rel", "--multi", multi, "--genome", gdef, "-o", out, sam])
    assert r.returncode == 0, r.stderr.decode()
    head, rows = read_profile(out)
    # expected order: the oracle's restatement of the reference's key walk (pinned to the reference's own
    # table by tests/test_genome_order_cpu.py)
    names = [gen_of[i].encode() for i in order_in_file]
    arr = (C.c_char_p * len(names))(*names)
    order = (C.c_int32 * len(names))()
    lib = orc.lib()
    lib.orc_key_order.restype = C.c_int32
    nk = lib.orc_key_order(arr, C.c_int32(len(names)), order)
    feats = [names[order[i]].decode() for i in range(nk)]
    assert nk == n_gen and feats != sorted(feats) and feats != list(dict.fromkeys(n.decode() for n in names))
    assert [row[0] for row in rows] == ["ID", "Unknown"] + feats
    # expected values
    fidx = {g: i for i, g in enumerate(feats)}
    fmap = np.array([fidx[g] for g in gen_of], dtype=np.int32)
    flen = np.zeros(n_gen, dtype=np.uint32)
    for i in range(n_seq):
        flen[fmap[i]] += lens[i]
    _, rec = samio.read_sam(sam)
    ref = orc.run_profile(rec, n_gen, multi=multi, fmap=fmap)
    vals, _, _ = orc.profile_finish(ref["abundance"], flen, ref["stats"], unit="rel", multi=multi)
    got = np.array([float(row[1]) for row in rows[1:]])
    assert got.shape == vals.shape
    assert (np.abs(got - vals) <= 1e-6 * np.maximum(np.abs(vals), 1e-12)).all(), (got, vals)


def test_cli_profile_genome_errors(tmp_path):
    """msam_profile.c:787,826,835: the three fatal messages of the genome definition (raised on the host,
    before any device work)."""
    sam = str(tmp_path / "g.sam")
    open(sam, "w").write("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:a\tLN:100\n@SQ\tSN:b\tLN:100\n")
    for text, msg in (("g1\n", b"GENOME DEFINITION LINE ERROR"), ("g1\ta\ng1\tzzz\n", b"Sequence 'zzz' not found in BAM file"),
                      ("g1\ta\n", b"Sequence 'b' not found in genome definition")):
        gdef = str(tmp_path / "d.tsv")
        open(gdef, "w").write(text)
        r = run(["profile", "-S", "--label", "t", "--genome", gdef, "-o", str(tmp_path / "o.gz"), sam])
        assert r.returncode == 1 and msg in r.stderr, (text, r.stderr)


# ---- the pipelined BAM reader (decode stage only: no GPU needed) ---------------------------------------
def _pipetest(path, mode, stats, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([DEV, "pipetest", str(mode), str(stats), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=e, timeout=300)
    return r.returncode, r.stdout.decode() + r.stderr.decode()


@pytest.fixture(scope="module")
def synth_bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipe")
    out = {}
    for flag in ("b", "u"):
        p = str(d / f"in_{flag}.bam")
        with open(p, "wb") as fh:
            subprocess.check_call([DEV, "synth", "--groups", "60000", "--refs", "500", f"-{flag}"], stdout=fh)
        out[flag] = p
    return out


@pytest.mark.parametrize("mode,stats", [(0, 1), (1, 1), (1, 0), (2, 0)])
def test_pipeline_decode_equals_record_reader(synth_bams, mode, stats):
    """Speculative parallel record chase, SoA packing, pool boundaries (msam_filter.c:120-125,170 /
    msam_profile.c:223-232) and the carry between batches against msh_read + the per-record rules: same
    records, same SoA fields, same pools -- for many small batches, few threads or many."""
    for flag, env in (("b", dict(MSX_BATCH_BYTES=2_000_000, MSX_THREADS=8)),
                      ("u", dict(MSX_BATCH_BYTES=300_000, MSX_BATCH_RECORDS=110_000, MSX_THREADS=5)),
                      ("u", dict(MSX_THREADS=3))):
        rc, text = _pipetest(synth_bams[flag], mode, stats, **env)
        assert rc == 0, text


def test_pipeline_decode_of_sam_text_equals_record_reader(synth_bams, tmp_path):
    """SAM text through the decode stage (msh_sam_append: chunks of lines parsed into BAM records on all threads) against
    the line-at-a-time reader: same records, same SoA fields, same pools, for chunks that cut lines anywhere."""
    sam = str(tmp_path / "in.sam")
    with open(sam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", synth_bams["u"]], stdout=fh)
    for mode, stats in ((1, 1), (2, 0)):
        for env in (dict(MSX_SAM_CHUNK=70_001, MSX_BATCH_BYTES=300_000, MSX_BATCH_RECORDS=110_000, MSX_THREADS=5),
                    dict(MSX_SAM_CHUNK=1_000_000, MSX_THREADS=16), dict(MSX_THREADS=1)):
            rc, text = _pipetest(sam, mode, stats, **env)
            assert rc == 0, text
    # CRLF line ends and a last line without a newline
    body = open(sam, "rb").read().replace(b"\n", b"\r\n")[:-2]
    open(sam, "wb").write(body)
    rc, text = _pipetest(sam, 1, 1, MSX_SAM_CHUNK=50_000, MSX_THREADS=4)
    assert rc == 0, text



@pytest.mark.parametrize("flag", ["b", "u"])
@pytest.mark.parametrize("blocks", [1, 7, 64, 4096])
def test_block_feed_of_the_device_inflater(synth_bams, flag, blocks):
    """msh_raw_append (block headers walked, DEFLATE payloads copied, table written) + msh_inflate_table, without a
    device (`msamtools rawtest`): the record stream they reproduce against Python's own decompression of the file."""
    import gzip
    import struct
    import zlib
    raw = gzip.open(synth_bams[flag], "rb").read()
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    out = subprocess.check_output([DEV, "rawtest", "--blocks", str(blocks), synth_bams[flag]],
                                  env=dict(os.environ, MSX_THREADS="5", MSX_INFLATE_BLOCKS="8")).decode().split()
    got = dict(kv.split("=") for kv in out)
    assert int(got["bytes"]) == len(raw) - p
    assert int(got["crc32"], 16) == zlib.crc32(raw[p:])
    assert int(got["batches"]) >= (2 if blocks < 64 else 1)

@pytest.mark.parametrize("flag", ["-u", "-b"])
def test_stream_writer_roundtrip(synth_bams, tmp_path, flag):
    """msh_write_stream -- the writer of device-unpacked batches: a ready-made record stream cut into BGZF payloads where
    they fall, records straddling blocks -- through a file and through a pipe (vmsplice), several batches: the records
    read back are the input's, in order (`msamtools digest`)."""
    want = subprocess.check_output([DEV, "digest", synth_bams["u"]])
    env = dict(os.environ, MSX_BATCH_BYTES="3000000", MSX_INFLATE_BLOCKS="16", MSX_THREADS="6")
    out = str(tmp_path / "o.bam")
    with open(out, "wb") as fh:
        subprocess.check_call([DEV, "restream", flag, synth_bams["b"]], stdout=fh, env=env)
    assert subprocess.check_output([DEV, "digest", out]) == want
    subprocess.check_call(f"{DEV} restream {flag} {synth_bams['u']} | cat > {out}", shell=True, env=env)
    assert subprocess.check_output([DEV, "digest", out]) == want
    # an independent reader agrees on the BGZF framing (python gzip + struct)
    hdr, rec = samio.read_bam(out)
    assert rec.n == int(want.split()[0].split(b"=")[1])


def test_pipeline_chase_repairs_wrong_guesses(synth_bams):
    """With MSX_CHASE_SLOPPY nearly every guessed record start is wrong: the stitcher must still
    deliver the true chain."""
    rc, text = _pipetest(synth_bams["u"], 1, 1, MSX_CHASE_SLOPPY=1, MSX_BATCH_BYTES=4_000_000, MSX_THREADS=16)
    assert rc == 0, text


def test_pipeline_long_unmapped_stretch_is_linear(tmp_path):
    """A name-grouped file with a long tail of unmapped records (the pool rule walked back
    from every record to the last mapped one, quadratic in the length of such a stretch)."""
    sam = tmp_path / "tail.sam"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:r1\tLN:1000\n")
        for i in range(2000):
            fh.write(f"m{i:06d}\t0\tr1\t10\t255\t10M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:10\tAS:i:10\n")
        for i in range(400_000):
            fh.write(f"u{i:07d}\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n")
    bam = tmp_path / "tail.bam"
    with open(bam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-u", str(sam)], stdout=fh)
    t0 = time.time()
    rc, text = _pipetest(str(bam), 1, 1, MSX_THREADS=4)
    assert rc == 0, text
    assert "pools=402000" in text            # every unmapped record with a new name closes a pool (msam_filter.c:120-125)
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_cli_profile_over_a_one_rank_communicator(tmp_path, synth_bams):
    """`profile` started as a rank (RCCL communicator, counts and the per-iteration increment all-reduced
    inside msx_profile_finalize_dist_enqueue) writes the same file as the single-process run; also through
    the "{rank}" shard path with WORLD_SIZE / RANK in the environment."""
    a, b, c = str(tmp_path / "a.gz"), str(tmp_path / "b.gz"), str(tmp_path / "c.gz")
    base = ["profile", "--label", "S", "--multi", "prop"]
    assert run(base + ["-o", a, synth_bams["b"]]).returncode == 0
    r = run(base + ["-o", b, synth_bams["b"]], env={"MSX_FORCE_DIST": "1", "MSX_CLEAN_EXIT": "1"})
    assert r.returncode == 0, r.stderr.decode()
    text = lambda p: "\n".join(l for l in gzip.open(p, "rt").read().split("\n") if not l.startswith("# Command"))
    assert text(a) == text(b)
    # MSX_DIST_SLICES: the per-iteration all-reduce in slices of the feature range, on a side stream -- the same file
    r = run(base + ["-o", b, synth_bams["b"]], env={"MSX_FORCE_DIST": "1", "MSX_CLEAN_EXIT": "1", "MSX_DIST_SLICES": "3"})
    assert r.returncode == 0, r.stderr.decode()
    assert text(a) == text(b)
    shard = str(tmp_path / "shard0.bam")
    os.link(synth_bams["b"], shard)
    r = run(base + ["-o", c, str(tmp_path / "shard{rank}.bam")],
            env={"MSX_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"})
    assert r.returncode != 0 or text(c) == text(a)      # ("{rank}" is substituted only for WORLD_SIZE > 1)


@pytest.mark.gpu
def test_cli_fatal_record_writes_what_precedes_it(tmp_path):
    """A record with neither MD nor NM is fatal for -p (msam_filter.c:150-152).  The reference dies AT that record,
    after having written the pools before it.  Without best-hit selection the batches carry no filter pools: the
    batch is filtered again up to the offending record, every earlier batch and those records are written, then the
    reference's message, exit 1 (with --besthit / --uniqhit the cut is in front of the record's pool: next test)."""
    sam = tmp_path / "bad.sam"
    good = "r{0}\t0\tchr1\t{1}\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:50\tAS:i:50\n"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:100000\n")
        for i in range(3):
            fh.write(good.format(i, 100 + i))
        fh.write("r3\t0\tchr1\t200\t255\t50M\t*\t0\t0\t*\t*\tAS:i:50\n")          # no MD, no NM
        fh.write(good.format(4, 300))
    r = run(["filter", "-S", "-p", "95", str(sam)])
    assert r.returncode == 1
    assert r.stderr.decode().strip().endswith(
        "Fatal Error: Either NM or MD must be present in SAM/BAM input for 'filter' command. "
        "Type 'msamtools filter -h' for details.")
    assert r.stdout.decode() == "".join(good.format(i, 100 + i) for i in range(3))     # r0..r2, as the reference
    # the same file through the BAM pipeline (BAM out: the records sit in the open block, which -- like htslib's last
    # buffer when the reference dies -- is not written)
    bam = tmp_path / "bad.bam"
    bam.write_bytes(run(["recode", "-b", str(sam)]).stdout)
    r = run(["filter", "-p", "95", str(bam)])
    assert r.returncode == 1 and b"Either NM or MD must be present" in r.stderr
    r = run(["filter", "-p", "95", "-h", str(bam)])
    assert r.returncode == 1


@pytest.mark.gpu
def test_cli_fatal_record_leaves_its_open_pool_unwritten(tmp_path):
    """Plain -l/-p/-z: the reference keeps the records of the current QNAME in an open pool and writes it at the next name
    change (msam_filter.c:120-125); the name it compares with is the last MAPPED record's (:170), an unmapped record does
    not move it on (:132-138).  A record without MD and NM kills it with that pool unwritten (:150-152).  The command line
    cuts the batch where that pool began -- against the oracle's output up to its error, SAM and BAM, device-side and
    host-side walk."""
    import oracle_lib as orc
    good = "{0}\t{1}\tchr1\t{2}\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:50\tAS:i:50\n"
    bad = "{0}\t{1}\tchr1\t{2}\t255\t50M\t*\t0\t0\t*\t*\tAS:i:50\n"
    unm = "{0}\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n"
    cases = {
        "same_name_in_front": [good.format("a", 0, 100), good.format("b", 0, 110), good.format("c", 0, 120), good.format("c", 16, 130),
                               bad.format("c", 0, 140), good.format("d", 0, 150)],
        "new_name": [good.format("a", 0, 100), good.format("b", 0, 110), bad.format("c", 0, 140), good.format("d", 0, 150)],
        "unmapped_in_between": [good.format("a", 0, 100), good.format("b", 0, 110), unm.format("b"), good.format("b", 0, 115),
                                unm.format("x"), bad.format("b", 0, 140)],
        "first_record": [bad.format("a", 0, 100), good.format("b", 0, 110)],
    }
    for label, lines in cases.items():
        sam = tmp_path / f"{label}.sam"
        sam.write_text("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:100000\n" + "".join(lines))
        _, rec = samio.read_sam(str(sam))
        want = orc.run_filter(rec, p=95)
        assert want["rc"] != 0
        expect = "".join(lines[i] for i in want["emit"])
        r = run(["filter", "-S", "-p", "95", str(sam)])
        assert r.returncode == 1 and b"Either NM or MD must be present" in r.stderr, (label, r.stderr)
        assert r.stdout.decode() == expect, (label, want["emit"])
        bam = tmp_path / f"{label}.bam"
        bam.write_bytes(run(["recode", "-b", str(sam)]).stdout)
        for env in ({}, {"MSX_HOST_UNPACK": "1"}):
            r = run(["filter", "-p", "95", str(bam)], env=env)
            assert r.returncode == 1 and b"Either NM or MD must be present" in r.stderr
            assert r.stdout.decode() == expect, (label, env)


@pytest.mark.gpu
def test_cli_fatal_record_after_the_pools_before_it(tmp_path, synth_bams):
    """With best-hit pools the reference has written every pool it completed when its loop dies at a record
    (msam_filter.c:150-152 no MD/NM, :219-221 no AS).  So does the command line: the batch is filtered again in front
    of the offending pool, every earlier batch and those pools are written, then the reference's message, exit 1."""
    sam = tmp_path / "bad.sam"
    good = "r{0}\t0\tchr1\t{1}\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:50\tAS:i:50\n"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:100000\n")
        for i in range(3):
            fh.write(good.format(i, 100 + i))
        fh.write("r3\t0\tchr1\t200\t255\t50M\t*\t0\t0\t*\t*\tAS:i:50\n")          # no MD, no NM
        fh.write(good.format(4, 300))
    r = run(["filter", "-S", "-p", "95", "--besthit", str(sam)])
    assert r.returncode == 1 and b"Either NM or MD must be present" in r.stderr
    assert [l.split("\t")[0] for l in r.stdout.decode().split("\n") if l] == ["r0", "r1", "r2"]
    # no AS on a participating record (the pool's writer dies): the pools before it are out
    text = open(sam).read().replace("r3\t0\tchr1\t200\t255\t50M\t*\t0\t0\t*\t*\tAS:i:50", "r3\t0\tchr1\t200\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:50")
    open(sam, "w").write(text)
    r = run(["filter", "-S", "--besthit", str(sam)])
    assert r.returncode == 1 and b"Required field AS not found" in r.stderr
    assert [l.split("\t")[0] for l in r.stdout.decode().split("\n") if l] == ["r0", "r1", "r2"]
    # many batches, the offending record late in the file: every earlier batch is written (in order), then the pools
    # of its own batch in front of it -- device-side walk and host-side walk alike
    src = subprocess.check_output([DEV, "recode", synth_bams["u"]]).decode().split("\n")[:-1]
    k = next(i for i in range(len(src) * 3 // 4, len(src)) if src[i].split("\t")[0] != src[i - 1].split("\t")[0])
    f = src[k].split("\t")
    src[k] = "\t".join(x for x in f if not x.startswith(("MD:", "NM:")))
    hdr = subprocess.check_output([DEV, "recode", "-h", synth_bams["u"]]).decode().split("\n")
    hdr = [l for l in hdr if l.startswith("@")]
    bad_sam = tmp_path / "late.sam"
    bad_sam.write_text("\n".join(hdr + src) + "\n")
    bad_bam = tmp_path / "late.bam"
    bad_bam.write_bytes(run(["recode", "-u", str(bad_sam)]).stdout)
    import oracle_lib as orc
    _, rec = samio.read_sam(str(bad_sam))
    want = orc.run_filter(rec, l=80, p=95, z=80, besthit=True)
    assert want["rc"] != 0 and want["err_record"] == k
    # what the oracle emits from the records in front of the offending one (its pool starts there: k is a name change)
    class Cut:
        pass
    cut = Cut()
    for a in ("flag", "rflags", "tid", "pos", "nm", "as_"):
        setattr(cut, a, getattr(rec, a)[:k])
    cut.cigar_off, cut.md_off, cut.cigar, cut.md = rec.cigar_off[:k + 1], rec.md_off[:k + 1], rec.cigar, rec.md
    cut.qname_off, cut.qname = rec.qname_off[:k + 1], rec.qname
    pre = orc.run_filter(cut, l=80, p=95, z=80, besthit=True)
    assert pre["rc"] == 0
    for env in ({}, {"MSX_HOST_UNPACK": "1"}):
        e = dict(MSX_BATCH_BYTES="1500000", MSX_BATCH_RECORDS="110000", MSX_INFLATE_BLOCKS="8", MSX_THREADS="8", **env)
        r = run(["filter", "-l", "80", "-p", "95", "-z", "80", "--besthit", str(bad_bam)], env=e)
        assert r.returncode == 1 and b"Either NM or MD must be present" in r.stderr
        got = r.stdout.decode().split("\n")[:-1]
        assert got == [src[i] for i in pre["emit"]], (len(got), len(pre["emit"]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--besthit", "--uniqhit"])
def test_cli_fatal_pool_written_as_far_as_the_reference_got(tmp_path, mode):
    """A paired pool is written READ1 pass first, then READ2 pass (msam_filter.c:247-263); a record without AS is met by the
    pass of its mate class (:219-221).  So when only READ2 records lack AS the reference has written the pool's READ1
    winners by the time it dies; when a READ1 record lacks AS, or the pool is unpaired, nothing of the pool.  Every
    arrangement against the oracle's output up to its error (it restates the writer pass by pass)."""
    import oracle_lib as orc
    rec_t = "{name}\t{flag}\tchr1\t{pos}\t255\t50M\t*\t0\t0\t*\t*\tNM:i:0\tMD:Z:50{as_}\n"
    def pool(name, rows):          # rows: (flag, AS or None)
        return [rec_t.format(name=name, flag=f, pos=100 + 10 * i, as_="" if a is None else f"\tAS:i:{a}") for i, (f, a) in enumerate(rows)]
    R1, R2 = 0x41, 0x81
    cases = {
        "read2_lacks_as": [(R1, 40), (R1, 50), (R2, 30), (R2, None), (R1, 50)],          # both READ1 winners out, then death
        "read2_lacks_as_first_in_pool": [(R2, None), (R1, 45), (R2, 60)],
        "read1_lacks_as": [(R1, 40), (R2, 50), (R1, None), (R2, 50)],                   # dies in the READ1 pass: nothing of the pool
        "both_lack_as": [(R2, None), (R1, None), (R1, 50)],
        "unpaired_lacks_as": [(0, 50), (0, None), (0, 50)],
        "read1_tie_read2_lacks_as": [(R1, 50), (R1, 50), (R2, None)],                   # --uniqhit: the tie writes nothing, then death
    }
    for label, rows in cases.items():
        sam = tmp_path / f"{label}.sam"
        lines = pool("a0", [(R1, 50), (R2, 50)]) + pool("a1", [(0, 50)]) + pool("bad", rows) + pool("z9", [(0, 50)])
        sam.write_text("@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:chr1\tLN:100000\n" + "".join(lines))
        _, rec = samio.read_sam(str(sam))
        want = orc.run_filter(rec, **{mode[2:]: True})
        assert want["rc"] != 0
        r = run(["filter", "-S", mode, str(sam)])
        assert r.returncode == 1 and b"Required field AS not found" in r.stderr, (label, r.stderr)
        assert r.stdout.decode() == "".join(lines[i] for i in want["emit"]), (label, mode, want["emit"])
        # the same through the BAM pipeline, device-side walk and host-side walk (SAM text out)
        bam = tmp_path / f"{label}.bam"
        bam.write_bytes(run(["recode", "-b", str(sam)]).stdout)
        for env in ({}, {"MSX_HOST_UNPACK": "1"}):
            r = run(["filter", mode, str(bam)], env=env)
            assert r.returncode == 1 and b"Required field AS not found" in r.stderr
            assert r.stdout.decode() == "".join(lines[i] for i in want["emit"]), (label, mode, env)


@pytest.mark.gpu
def test_cli_coverage_binned_path_equals_atomic_path(tmp_path, synth_bams):
    """The binned pile-up (items sorted by tile, LDS image per tile) is what large batches take; forced on a
    small input (MSX_COV_BINNED_FROM=1) it must write the same per-base depths as the atomic path, and the
    reference's golden coverage table still holds through it (msam_coverage.c:33-87)."""
    a, b = str(tmp_path / "a.gz"), str(tmp_path / "b.gz")
    assert run(["coverage", "-o", a, synth_bams["b"]], env={"MSX_COV_BINNED_FROM": "1000000000"}).returncode == 0
    r = run(["coverage", "-o", b, synth_bams["b"]], env={"MSX_COV_BINNED_FROM": "1"})
    assert r.returncode == 0, r.stderr.decode()
    assert gzip.open(a, "rb").read() == gzip.open(b, "rb").read()
    blk = EXP["coverage"]
    out = str(tmp_path / "positions.gz")
    r = run(["coverage", "-S", "-w", "4", "-o", out, fixture_path(blk["fixture"])], env={"MSX_COV_BINNED_FROM": "1"})
    assert r.returncode == 0
    want = []
    for name in ("A", "B", "C", "D"):
        v = blk["positions"][name]
        want.append(">" + name)
        for i in range(0, len(v), 4):
            want.append(" ".join(str(x) for x in v[i:i + 4]))
    assert gzip.open(out, "rt").read().split("\n")[:-1] == want


def test_product_binary_carries_only_the_reference_commands():
    """generator and I/O self-tests live in msamtools-dev (msh_dev.c); `msamtools` dispatches what msamtools.c:31-49 does"""
    for c in ("synth", "recode", "digest", "pipetest", "restream", "rawtest", "keyorder"):
        r = subprocess.run([BIN, c], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"unrecognized command" in r.stderr, c
    assert subprocess.run([DEV, "filter"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 1


# ---- `summary` (msam_summary.c) ----------------------------------------------------------------------------------------

def test_summary_count_and_option_errors_need_no_gpu():
    """--count reads names and flags only (mCountInserts, msam_summary.c:19-40); the option conflicts are refused before the
    input is opened, with the reference's lines on stdout (:222-242), an unknown --stats mode with its mDie text (:266)."""
    for case in EXP["summary"]["cases"]:
        if "-c" in case["args"]:
            r = run(["summary", "-S"] + case["args"] + [fixture_path(case["fixture"])])
            assert r.returncode == 0 and r.stdout.decode().split("\n")[:-1] == case["stdout"], case["src"]
            assert r.stderr == b""
    fx = fixture_path("summary.sam")
    r = run(["summary", "-S", "-e", "-1", fx])
    assert r.returncode == 1 and r.stdout.decode().startswith("-e must be a positive integer\n")
    r = run(["summary", "-S", "--stats", "edit", "-c", fx])
    assert r.returncode == 1 and r.stdout.decode().startswith("--stats cannot be combined with --count\n")
    r = run(["summary", "-S", "-e", "3", "-c", fx])
    assert r.returncode == 1 and r.stdout.decode().startswith("-e cannot be combined with --count\n")
    r = run(["summary", "-S", "--stats", "median", fx])
    assert r.returncode == 1 and r.stderr.decode().strip().endswith("Fatal Error: Do not understand median as mode")
    r = run(["summary", "--help"])
    assert r.returncode == 0 and b"qname,aligned_qlen,target_name,glocal_align_len,matches,percent_identity" in r.stdout
    assert b"summary        summarize alignment statistics per read in a table format" in run(["help"]).stdout


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXP["summary"]["cases"], ids=[c["name"] for c in EXP["summary"]["cases"]])
def test_cli_summary_golden(case):
    """test_summary.sh through the command line: the table, --edge, the four --stats distributions, --count, M against =/X."""
    r = run(["summary", "-S"] + case["args"] + [fixture_path(case["fixture"])])
    assert r.returncode == 0, r.stderr.decode()
    got = r.stdout.decode().split("\n")[:-1]
    if "stdout" in case:
        assert got == case["stdout"], case["src"]
    for x in case.get("contains", []):
        assert any(x in l for l in got), case["src"]
    for x in case.get("not_contains", []):
        assert not any(x in l for l in got), case["src"]
    if case.get("stderr_empty"):
        assert r.stderr == b""


@pytest.mark.gpu
def test_cli_summary_equals_the_oracle(tmp_path, synth_bams):
    """Every fixture (records without MD, with NM only, soft and hard clips, I / D / N, unmapped and secondary ones) and a
    300 000-record BAM through `summary` -- table, --edge, every --stats mode, --count -- against the oracle's restatement of
    msam_summary.c and bam_get_extended_summary (mBamVector.c:135-236); BAM input takes the bulk reader, SAM the text one."""
    import oracle_lib as orc
    sams = [fixture_path(f) for f in ("summary.sam", "summary_count.sam", "summary_edge.sam", "filter.sam", "cigar_eqx.sam", "besthit.sam",
                                      "profile.sam", "integration.sam", "coverage.sam", "long_qname.sam")]
    big_sam = tmp_path / "big.sam"
    big_sam.write_bytes(run(["recode", "-h", synth_bams["b"]]).stdout)
    for path in sams + [str(big_sam)]:
        hdr, rec = samio.read_sam(path)
        names = [rec.name(i) for i in range(rec.n)]
        inputs = [path]
        if path == str(big_sam):
            inputs.append(synth_bams["b"])                     # the same records as BAM
        for inp in inputs:
            for edge in (0, 7):
                e = ["-e", str(edge)] if edge else []
                r = run(["summary"] + e + [inp])
                assert r.returncode == 0, (inp, r.stderr.decode()[-500:])
                assert r.stdout.decode().split("\n")[:-1] == orc.summary_lines(rec, names, hdr.target_name, hdr.target_len, edge), (inp, edge)
                for which in ("mapped", "unmapped", "edit", "score"):
                    r = run(["summary", "--stats", which] + e + [inp])
                    assert r.returncode == 0
                    assert r.stdout.decode().split("\n")[:-1] == orc.summary_stats(rec, hdr.target_len, which, edge), (inp, which, edge)
            r = run(["summary", "-c", inp])
            assert r.returncode == 0 and r.stdout.decode().strip() == str(orc.summary_count(rec)), inp
