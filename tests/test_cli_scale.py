"""The command line at scale against the oracle (-m gpu): the assembled decode | device | encode pipeline --
BGZF inflate, record chase, SoA packing, pools carried across batch cuts, kernels, emit-index gather into BGZF
blocks, the writer thread, the pipe between two processes -- on a synthetic BAM of millions of records cut into
twenty and more batches, 16 host threads.

What is compared (msam_filter.c:206-245, msam_profile.c:858-983; the comparison rule of the reference's own
harness, tests/functions.sh:160-163: same records in the same order, same profile):
  * `filter -l 80 -p 95 -z 80 --besthit` output (-bu, -b, file and pipe): the records the oracle emits, in its
    order -- as text lines (`recode`) on the 1 M-record file, as the order-sensitive digest of (QNAME, FLAG, tid,
    pos) on the 3 M-record file (tests/digest.py = `msamtools digest`);
  * `filter ... | profile -`, `profile` of filter's file and `profile` alone: header counts equal the oracle's,
    every value within 1e-6 relative of orc.run_profile + profile_finish.
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

import digest
import oracle_lib as orc
from conftest import ROOT

BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)
FILT = ["filter", "-l", "80", "-p", "95", "-z", "80", "--besthit"]
OPTS = dict(l=80, p=95, z=80, besthit=True)
REF_LEN = 4496          # msh_dev.c: synth_main writes every @SQ with this length
ENV = dict(MSX_THREADS="16", MSX_BATCH_BYTES="2500000", MSX_BATCH_RECORDS="160000", MSX_INFLATE_BLOCKS="24")

pytestmark = pytest.mark.gpu


def sh(cmd, **env):
    e = dict(os.environ, **ENV)
    e.update({k: str(v) for k, v in env.items()})
    # (a command line that hangs must fail its test, not hold the GPU box until the harness gives up: round 4 lost forty
    #  minutes to a deadlock between device threads over output buffers)
    r = subprocess.run(f"timeout 180 {cmd}" if "|" not in cmd else f"timeout 180 bash -c {cmd!r}", shell=True, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (cmd, r.returncode, r.stderr.decode()[-2000:])
    return r


class Case:
    """A synthetic name-grouped BAM (-b and -u) with the oracle's answers for it."""

    def __init__(self, d, groups, refs):
        import msamtools_amd as m
        self.groups, self.refs, self.dir = groups, refs, str(d)
        self.bam = {}
        for flag in ("b", "u"):
            p = os.path.join(self.dir, f"in_{groups}_{flag}.bam")
            with open(p, "wb") as fh:
                subprocess.check_call([DEV, "synth", "--groups", str(groups), "--refs", str(refs), f"-{flag}"], stdout=fh)
            self.bam[flag] = p
        self.hs = m.HostSynth(13579, groups, refs, 4)
        f = orc.run_filter(self.hs, **OPTS)
        assert f["rc"] == 0
        self.emit = f["emit"]
        self.digest_in = digest.synth_digest(self.hs)
        self.digest_out = digest.synth_digest(self.hs, self.emit)
        self.full_out = {}
        self.flen = np.full(refs, REF_LEN, dtype=np.uint32)
        self.pipe = orc.run_profile(self.hs, refs, multi="proportional", sel=self.emit)
        self.plain = orc.run_profile(self.hs, refs, multi="proportional")

    def check_digest(self, path, want, src=None):
        env = dict(os.environ, MSX_THREADS="8")
        out = subprocess.check_output([DEV, "digest", path], env=env).decode().strip()
        h, n = want
        assert out == f"records={n} digest={h:016x}", (path, out, n, f"{h:016x}")
        if want is self.digest_out:
            # filter's output holds the selected input records byte for byte (msam_filter.c:232-244 writes the pooled
            # bam1_t as read): the digest over EVERY byte of every record, against the input's records at the oracle's
            # emit list (msamtools-dev digest --full --select; tests/test_digest_cpu.py)
            src = src or self.bam["u"]
            if src not in self.full_out:
                idx = os.path.join(self.dir, "emit.u32")
                np.asarray(self.emit).astype("<u4").tofile(idx)
                self.full_out[src] = subprocess.check_output([DEV, "digest", "--full", "--select", idx, src], env=env).decode().strip()
                assert self.full_out[src].startswith(f"records={n} ")
            assert subprocess.check_output([DEV, "digest", "--full", path], env=env).decode().strip() == self.full_out[src], path

    def check_profile(self, path, ref, batches_stderr=None):
        text = gzip.open(path, "rt").read()
        head = [l for l in text.split("\n") if l.startswith("#")]
        rows = [l.split("\t") for l in text.split("\n") if l and not l.startswith("#")]
        s = ref["stats"]
        get = lambda key: next(l for l in head if l.startswith(key)).split(":")[1].split("(")[0].strip()
        assert int(get("# Mapped inserts")) == s.insert_count
        assert int(get("#   - Multiple mapped")) == s.multi_mapper_count
        assert int(get("#   - Uniquely mapped")) == s.uniq_mapper_count
        vals, purged, eff = orc.profile_finish(ref["abundance"], self.flen, s, unit="rel")
        assert float(get("# Purged inserts")) == pytest.approx(purged, rel=1e-6)
        assert float(get("# Effective inserts")) == pytest.approx(eff, rel=1e-6)
        assert rows[0] == ["ID", "S"] and rows[1][0] == "Unknown" and len(rows) == self.refs + 2
        assert [r[0] for r in rows[2:5]] == ["ref0000000", "ref0000001", "ref0000002"]
        got = np.array([float(r[1]) for r in rows[1:]])
        assert np.array_equal(got == 0, vals == 0)
        # %.8g prints eight significant digits: 5e-8 relative on top of the 1e-6 the profile is held to
        rel = np.abs(got - vals) / np.maximum(np.abs(vals), 1e-300)
        assert rel.max() <= 1e-6 + 1e-7, rel.max()
        assert abs(got.sum() - 1.0) <= 5e-6


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    return Case(tmp_path_factory.mktemp("scale3m"), 600_000, 3000)


@pytest.fixture(scope="module")
def mid(tmp_path_factory):
    return Case(tmp_path_factory.mktemp("scale1m"), 200_000, 800)


def n_batches(stderr):
    for line in stderr.decode().split("\n"):
        if line.startswith("# batches:"):
            return int(line.split()[2].rstrip(";"))
    return None


def test_inputs_are_what_the_oracle_was_given(big, mid):
    for c in (big, mid):
        for flag in ("b", "u"):
            c.check_digest(c.bam[flag], c.digest_in)


def test_filter_records_as_text_lines(mid, tmp_path):
    """1 M records, > 20 batches: every output line equals the input line the oracle selects, in its order."""
    out = str(tmp_path / "f.bam")
    r = sh(f"{BIN} {' '.join(FILT)} -bu {mid.bam['u']} > {out}", MSX_TIMING=1, MSX_BATCH_BYTES=1_500_000,
           MSX_BATCH_RECORDS=110_000)
    assert n_batches(r.stderr) >= 8, r.stderr.decode()[-800:]
    src = subprocess.check_output([DEV, "recode", mid.bam["u"]]).decode().split("\n")[:-1]
    got = subprocess.check_output([DEV, "recode", out]).decode().split("\n")[:-1]
    assert len(src) == mid.hs.n_records
    assert len(got) == len(mid.emit)
    want = [src[i] for i in mid.emit]
    assert got == want
    # SAM text out of the same run equals the BAM's records
    r = sh(f"{BIN} {' '.join(FILT)} {mid.bam['b']} > {tmp_path / 'f.sam'}")
    assert open(tmp_path / "f.sam").read().split("\n")[:-1] == want


@pytest.mark.parametrize("inflag,outflag", [("b", "-bu"), ("u", "-bu"), ("b", "-b"), ("u", "-b")])
def test_filter_digest_many_batches(big, tmp_path, inflag, outflag):
    out = str(tmp_path / "f.bam")
    r = sh(f"{BIN} {' '.join(FILT)} {outflag} {big.bam[inflag]} > {out}", MSX_TIMING=1)
    assert n_batches(r.stderr) >= 20, r.stderr.decode()[-800:]
    big.check_digest(out, big.digest_out)


@pytest.mark.parametrize("env", [dict(MSX_INFLATE_REFUSE=7), dict(MSX_INFLATE_REFUSE=1), dict(MSX_HOST_INFLATE=1),
                                 dict(MSX_NO_INFLATE_AHEAD=1), dict(MSX_INFLATE_AHEAD=2), dict(MSX_INFLATE_WAVES=3),
                                 dict(MSX_UP_HEAD=256), dict(MSX_UP_HEAD=256, MSX_INFLATE_AHEAD=2)])
def test_where_the_blocks_are_inflated_changes_nothing(big, tmp_path, env):
    """BGZF blocks are inflated on the device (msx_inflate.hip); batches with a block the device refuses (here: every 7th
    block, by a test switch) are inflated on the host instead; MSX_HOST_INFLATE=1 inflates everything there.  Same output."""
    out, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    if "MSX_INFLATE_REFUSE" in env or "MSX_UP_HEAD" in env:
        # test hooks: only the debug build of the library knows them (msamtools_amd/dbg, -DMSX_DEBUG_SWITCHES); the product
        # binary finds its library through a RUNPATH, which LD_LIBRARY_PATH precedes.  (MSX_UP_HEAD: a batch sent ahead is
        # walked where it was inflated, the carry copied into the room in front of it; with 256 bytes of room most carries
        # do not fit and the batch is copied behind the carry instead, as until round 6)
        env = dict(env, LD_LIBRARY_PATH=os.path.join(ROOT, "msamtools_amd", "dbg") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S {big.bam['b']} > {out}", MSX_TIMING=1, **env)
    assert n_batches(r.stderr) >= 20
    err = r.stderr.decode()
    if "MSX_INFLATE_REFUSE" in env:
        assert "inflated on the host (blocks the device refused)" in err
        # every batch holds refused blocks: after the first few the device is no longer asked (the slots in flight still are)
        tried = int(err.split(" batches inflated on the host")[0].split("# ")[-1])
        assert 3 <= tried <= 12, tried
    elif "MSX_HOST_INFLATE" in env:
        assert "inflated on the device" not in err
    else:
        assert "inflated on the device" in err and "refused" not in err
    big.check_digest(out, big.digest_out)
    big.check_profile(p, big.pipe)
    r = sh(f"{BIN} profile --label S -o {p} {big.bam['b']}", MSX_TIMING=1, **env)
    big.check_profile(p, big.plain)


def test_batches_that_grow_once_the_command_is_under_way(big, tmp_path):
    """filter -b over a long input takes batches of twice the blocks from its eighth batch on, in larger buffers the pin thread
    makes while the command runs and the decode stage swaps in when it next holds the slot (msh_pipeline.c: comp_ramp, phase 3
    of pin_thread).  Here the same at a small scale (MSX_COMP_RAMP_FROM): 16-block batches, 32 from the third on -- fewer
    batches, the same records, the same profile."""
    out, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    base = dict(MSX_COMP_BLOCKS=16, MSX_COMP_BYTES=2 << 20, MSX_TIMING=1)
    r0 = sh(f"{BIN} {' '.join(FILT)} -b --profile-out {p} --label S {big.bam['b']} > {out}", MSX_COMP_RAMP_FROM=0, **base)
    big.check_digest(out, big.digest_out)
    r1 = sh(f"{BIN} {' '.join(FILT)} -b --profile-out {p} --label S {big.bam['b']} > {out}", MSX_COMP_RAMP_FROM=3, **base)
    big.check_digest(out, big.digest_out)
    big.check_profile(p, big.pipe)
    assert n_batches(r0.stderr) >= 20 and n_batches(r1.stderr) < n_batches(r0.stderr) * 0.8, (n_batches(r0.stderr), n_batches(r1.stderr))
    # -bu never ramps (its output buffers would have to grow with the batches)
    r2 = sh(f"{BIN} {' '.join(FILT)} -bu {big.bam['b']} > {out}", MSX_COMP_RAMP_FROM=3, **base)
    big.check_digest(out, big.digest_out)
    assert n_batches(r2.stderr) == n_batches(sh(f"{BIN} {' '.join(FILT)} -bu {big.bam['b']} > {out}", MSX_COMP_RAMP_FROM=0, **base).stderr)


def test_coverage_through_the_pipeline(mid, tmp_path):
    """`coverage` on a many-batch BAM: the pipeline of filter and profile (batches as compressed blocks, inflated and
    walked on the device) against the oracle's pile-up (msam_coverage.c:33-87), and against the serial reader."""
    cov = orc.coverage(mid.hs, [REF_LEN] * mid.refs)
    want = []
    for t, c in enumerate(cov):
        if not c.any() and not ((mid.hs.tid == t).any()):
            want.append(f"ref{t:07d}\t0\t0")
        else:
            want.append(f"ref{t:07d}\t{(c != 0).sum() / REF_LEN:.8f}\t{c.sum() / REF_LEN:.2f}")
    outs = {}
    for name, env in (("device", {}), ("host inflate", dict(MSX_HOST_INFLATE=1)), ("host walk", dict(MSX_HOST_UNPACK=1)),
                      ("serial", dict(MSX_SERIAL_IO=1))):
        o = str(tmp_path / "c.gz")
        sh(f"{BIN} coverage --summary -o {o} {mid.bam['b']}", MSX_BATCH_BYTES=1_500_000, MSX_BATCH_RECORDS=110_000, **env)
        outs[name] = gzip.open(o, "rt").read().split("\n")[:-1]
    assert outs["device"] == want
    for name in outs:
        assert outs[name] == outs["device"], name
    # per-position text of a few targets
    o = str(tmp_path / "t.gz")
    sh(f"{BIN} coverage -w 17 -o {o} {mid.bam['u']}", MSX_BATCH_BYTES=1_500_000, MSX_BATCH_RECORDS=110_000)
    text = gzip.open(o, "rt").read()
    blocks = text.split(">")[1:]
    assert len(blocks) == mid.refs
    for t in (0, 1, mid.refs // 2, mid.refs - 1):
        head, body = blocks[t].split("\n", 1)
        assert head == f"ref{t:07d}"
        assert np.array_equal(np.array(body.split(), dtype=np.int64), cov[t].astype(np.int64))


def test_filter_through_pipes(big, tmp_path):
    """stdin from a pipe (the reader thread's ring), stdout into a pipe (vmsplice hand-over)."""
    out = str(tmp_path / "f.bam")
    sh(f"cat {big.bam['b']} | {BIN} {' '.join(FILT)} -bu - | cat > {out}")
    big.check_digest(out, big.digest_out)


@pytest.mark.parametrize("inflag", ["b", "u"])
def test_filter_pipe_profile(big, tmp_path, inflag):
    """The reference's two-process workflow, filter's pools and profile's pools both cut across > 20 batches."""
    p = str(tmp_path / "p.gz")
    sh(f"{BIN} {' '.join(FILT)} -bu {big.bam[inflag]} | {BIN} profile --label S -o {p} -")
    big.check_profile(p, big.pipe)


def test_filter_file_then_profile(big, tmp_path):
    f, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    sh(f"{BIN} {' '.join(FILT)} -b {big.bam['u']} > {f}")
    r = sh(f"{BIN} profile --label S -o {p} {f}", MSX_TIMING=1)
    assert n_batches(r.stderr) >= 8
    big.check_profile(p, big.pipe)


@pytest.mark.parametrize("inflag", ["b", "u"])
def test_profile_alone(big, tmp_path, inflag):
    p = str(tmp_path / "p.gz")
    r = sh(f"{BIN} profile --label S -o {p} {big.bam[inflag]}", MSX_TIMING=1)
    assert n_batches(r.stderr) >= 20
    big.check_profile(p, big.plain)


def test_one_batch_equals_many(mid, tmp_path):
    """The default batch size (one batch here) and 1 or 3 threads write the same records as the many-batch run
    (BGZF block boundaries follow the batches and threads, so the decoded streams are compared)."""
    def records(**env):
        out = str(tmp_path / "o.bam")
        sh(f"{BIN} {' '.join(FILT)} -bu {mid.bam['b']} > {out}", **env)
        return subprocess.check_output([DEV, "recode", "-h", out])
    ref = records()
    for env in (dict(MSX_BATCH_BYTES=96 << 20, MSX_BATCH_RECORDS=3 << 20), dict(MSX_THREADS=1), dict(MSX_THREADS=3)):
        assert records(**env) == ref


# ---- one process instead of two: filter --profile-out ------------------------------------------------------------

@pytest.mark.parametrize("inflag", ["b", "u"])
def test_tee_equals_the_two_process_pipe(big, tmp_path, inflag):
    """`filter ... --profile-out p.gz --label S` = `filter ... | profile -` without the pipe, the second decode and
    the second process: same records on stdout, same profile (msx_filter_profile_enqueue per batch, > 20 batches)."""
    f, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    r = sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S {big.bam[inflag]} > {f}", MSX_TIMING=1)
    assert n_batches(r.stderr) >= 20
    big.check_digest(f, big.digest_out)
    big.check_profile(p, big.pipe)
    assert b"PropSharing Iteration" in r.stderr and b"# Purged " in r.stderr


def test_tee_without_best_hit_and_with_uniqhit(mid, tmp_path):
    """-l/-p/-z only (pools do not shape filter's output: the host hands over profile's pools over the records
    filter can write) and --uniqhit, against the oracle run as the two commands."""
    for opts, cli in ((dict(l=80, p=95, z=80), "-l 80 -p 95 -z 80"), (dict(uniqhit=True), "--uniqhit"),
                      (dict(l=80, p=97, besthit=True), "-l 80 -p 97 --besthit")):
        f = orc.run_filter(mid.hs, **opts)
        ref = orc.run_profile(mid.hs, mid.refs, multi="proportional", sel=f["emit"])
        fb, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
        sh(f"{BIN} filter {cli} -bu --profile-out {p} --label S {mid.bam['b']} > {fb}", MSX_BATCH_BYTES=1_500_000,
           MSX_BATCH_RECORDS=110_000)
        mid.check_digest(fb, digest.synth_digest(mid.hs, f["emit"]))
        mid.check_profile(p, ref)


def test_tee_options_and_refusals(mid, tmp_path):
    p = str(tmp_path / "p.gz")
    r = subprocess.run(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} {mid.bam['b']}", shell=True, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"--profile-out requires --label" in r.stdout
    r = subprocess.run(f"{BIN} {' '.join(FILT)} -bu --label S {mid.bam['b']}", shell=True, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"need --profile-out" in r.stdout
    # profile's own options travel: --multi all --unit ab --nolen --total
    f = orc.run_filter(mid.hs, **OPTS)
    ref = orc.run_profile(mid.hs, mid.refs, multi="all", sel=f["emit"])
    sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S --multi all --unit ab --nolen --total 400000 {mid.bam['b']} > /dev/null")
    vals, _, _ = orc.profile_finish(ref["abundance"], mid.flen, ref["stats"], unit="ab", nolen=True, total=400000, multi="all")
    rows = [l.split("\t") for l in gzip.open(p, "rt").read().split("\n") if l and not l.startswith("#")]
    got = np.array([float(x[1]) for x in rows[1:]])
    assert np.allclose(got, vals, rtol=1e-6, atol=0)


# ---- one process, several contexts (MSX_DEVICES): here two contexts on the one GPU -----------------------------------

def test_two_contexts_filter_profile_and_tee(big, tmp_path):
    """The decode stage deals the batches to two device threads (each its own context, stage, unpacker and profile), the
    writer puts filter's output back in input order, the profiles are merged on the first context before the sharing
    iterations (msx_profile_merge): same files as with one context.  Every context inflates and walks its own batches
    on the device; the stream's carry (open pool, cut record, last naming QNAME) travels from the context that finished
    batch k to the one that walks batch k + 1 (msx_unpack_carry -> msx_unpack_seed)."""
    f, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    r = sh(f"{BIN} {' '.join(FILT)} -bu {big.bam['b']} > {f}", MSX_DEVICES="0,0", MSX_TIMING=1)
    assert b"2 devices" in r.stderr and b"BGZF blocks inflated on the device" in r.stderr
    big.check_digest(f, big.digest_out)
    r = sh(f"{BIN} profile --label S -o {p} {big.bam['u']}", MSX_DEVICES="0,0", MSX_TIMING=1)
    assert b"2 devices" in r.stderr and b"BGZF blocks inflated on the device" in r.stderr
    big.check_profile(p, big.plain)
    r = sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S {big.bam['b']} > {f}", MSX_DEVICES="0,0,0", MSX_TIMING=1)
    assert b"3 devices" in r.stderr and b"BGZF blocks inflated on the device" in r.stderr
    big.check_digest(f, big.digest_out)
    big.check_profile(p, big.pipe)
    # compressed output (the device DEFLATE encoder on every context), and round 3's form: inflate and walk on the host
    sh(f"{BIN} {' '.join(FILT)} -b --profile-out {p} --label S {big.bam['b']} > {f}", MSX_DEVICES="0,0")
    big.check_digest(f, big.digest_out)
    big.check_profile(p, big.pipe)
    r = sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S {big.bam['b']} > {f}", MSX_DEVICES="0,0", MSX_MULTI_HOST_WALK=1, MSX_TIMING=1)
    assert b"BGZF blocks inflated on the device" not in r.stderr
    big.check_digest(f, big.digest_out)
    big.check_profile(p, big.pipe)
    # --multi equal: the contexts' integer shares are added as integers (msx_profile_merge) -- every digit one context prints
    one, three = str(tmp_path / "eq1.gz"), str(tmp_path / "eq3.gz")
    sh(f"{BIN} profile --multi equal --label S -o {one} {big.bam['u']}")
    sh(f"{BIN} profile --multi equal --label S -o {three} {big.bam['u']}", MSX_DEVICES="0,0,0", MSX_BATCH_BYTES=1_200_000)
    rows = lambda path: [l for l in gzip.open(path, "rt").read().split("\n") if not l.startswith("# Command")]
    assert rows(one) == rows(three)
    # device threads finish batches in any order while the writer takes them in input order: again and again, with few
    # buffers to go round.  Three hangs were found here: output buffers handed out first come, first served (then by batch
    # number, but to a later batch of the same number first); a thread waiting for its next batch with the writer's next
    # one still in its hands (-b, the encoder beside the next batch); every context joining the same helper threads.
    for rep in range(12):
        sh(f"{BIN} {' '.join(FILT)} {'-b' if rep % 2 else '-bu'} --profile-out {p} --label S {big.bam['b']} > {f}",
           MSX_DEVICES=("0,0", "0,0,0", "0,0,0,0")[rep % 3], MSX_BATCH_BYTES=1_200_000)
        big.check_digest(f, big.digest_out)
    big.check_profile(p, big.pipe)


# ---- --rescore on the pipeline ------------------------------------------------------------------------------------

def test_rescore_on_the_pipeline(mid, tmp_path):
    """--rescore (msam_filter.c:160-168: AS recomputed from MD/NM, the old AS dropped, AS:i appended) used to take the
    one-batch-at-a-time loop; the pipeline's writer now rewrites the emitted records in parallel.  Same text as the
    serial loop, and the AS values are the oracle's."""
    cmd = f"{BIN} filter -l 80 -p 95 -z 80 --rescore --besthit -bu {mid.bam['b']}"
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    sh(f"{cmd} > {a}", MSX_BATCH_BYTES=1_500_000, MSX_BATCH_RECORDS=110_000)
    sh(f"{cmd} > {b}", MSX_SERIAL_IO=1, MSX_BATCH_RECORDS=100_000)
    ta = subprocess.check_output([DEV, "recode", a]).decode().split("\n")[:-1]
    tb = subprocess.check_output([DEV, "recode", b]).decode().split("\n")[:-1]
    assert ta == tb
    f = orc.run_filter(mid.hs, l=80, p=95, z=80, rescore=True, besthit=True)
    assert len(ta) == len(f["emit"])
    got_as = [int(next(x for x in l.split("\t")[11:] if x.startswith("AS:i:"))[5:]) for l in ta]
    assert got_as == f["as_out"][f["emit"]].tolist()
    assert all(l.split("\t")[-1].startswith("AS:i:") for l in ta[:1000])          # appended at the end (:167)


# ---- the record walk on the device (msx_unpack) against the host-side walk ----------------------------------------

def test_device_unpack_equals_host_unpack(big, tmp_path):
    """From the second batch on the pipeline uploads the inflated bytes and the device finds the records, scans the aux
    blocks, packs the SoA arrays, compares QNAMEs and cuts the batch (msx_unpack); MSX_HOST_UNPACK=1 keeps the host-side
    walk for every batch.  Same records out (both equal the oracle's), same profile; > 20 batches each."""
    for env in (dict(), dict(MSX_HOST_UNPACK=1)):
        f, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
        r = sh(f"{BIN} {' '.join(FILT)} -bu --profile-out {p} --label S {big.bam['b']} > {f}", MSX_TIMING=1, **env)
        assert n_batches(r.stderr) >= 20
        big.check_digest(f, big.digest_out)
        big.check_profile(p, big.pipe)
        r = sh(f"{BIN} profile --label S -o {p} {big.bam['b']}", MSX_TIMING=1, **env)
        assert n_batches(r.stderr) >= 20
        big.check_profile(p, big.plain)
    # compressed output takes the same stream writer (payloads cut where they fall, deflated in parallel)
    f = str(tmp_path / "fb.bam")
    sh(f"{BIN} {' '.join(FILT)} -b {big.bam['u']} > {f}")
    big.check_digest(f, big.digest_out)


# ---- SAM text in (the reference's validation harness feeds .sam: validate_profiles.py:735-751) ----------------------

def test_sam_text_input_takes_the_pipeline(mid, tmp_path):
    """SAM text is parsed into BAM records on all threads by the decode stage and takes the same pipeline (device
    unpack from the second batch on): same records out, same profiles -- filter, profile, the one-process pipe, and
    text out of text in."""
    sam = str(tmp_path / "in.sam")
    with open(sam, "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", mid.bam["b"]], stdout=fh)
    assert os.path.getsize(sam) > 50_000_000
    f, p = str(tmp_path / "f.bam"), str(tmp_path / "p.gz")
    os.environ["MSX_SAM_CHUNK"] = "1000000"
    r = sh(f"{BIN} {' '.join(FILT)} -S -bu --profile-out {p} --label S {sam} > {f}", MSX_TIMING=1, MSX_BATCH_BYTES=4_000_000)
    assert n_batches(r.stderr) >= 5
    mid.check_digest(f, mid.digest_out, src=sam)        # (records as the line-at-a-time reader parses them: integer aux types by value)
    mid.check_profile(p, mid.pipe)
    r = sh(f"cat {sam} | {BIN} profile -S --label S -o {p} -", MSX_TIMING=1, MSX_BATCH_BYTES=4_000_000)
    assert n_batches(r.stderr) >= 5
    mid.check_profile(p, mid.plain)
    # text in, text out: the lines the oracle selects
    out = sh(f"{BIN} {' '.join(FILT)} -S {sam}", MSX_BATCH_BYTES=4_000_000).stdout.decode().split("\n")[:-1]
    src = [l for l in open(sam).read().split("\n") if l and not l.startswith("@")]
    assert out == [src[i] for i in mid.emit]
    # and the record-at-a-time reader agrees (MSX_SERIAL_IO)
    assert sh(f"{BIN} {' '.join(FILT)} -S {sam}", MSX_SERIAL_IO=1).stdout.decode().split("\n")[:-1] == out
    del os.environ["MSX_SAM_CHUNK"]


# ---- damaged record streams through the device-side walk -----------------------------------------------------------

def bgzf_blocks(data, level=1):
    """BGZF framing (SAMv1 4.1) of a byte string, in Python: blocks of <= 0xff00 payload bytes + the EOF block"""
    import struct
    import zlib
    out = bytearray()
    for i in range(0, len(data), 0xff00):
        chunk = data[i:i + 0xff00]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    return bytes(out)


def test_truncated_and_corrupt_record_streams_are_fatal(mid, tmp_path):
    """A BAM whose record stream ends inside a record, or holds a block_size < 32, dies with the host reader's own
    diagnostics when the batch is walked on the device as well (several batches, so the damage lies in a raw one)."""
    import gzip
    import struct
    raw = gzip.open(mid.bam["u"], "rb").read()
    cut = tmp_path / "cut.bam"
    cut.write_bytes(bgzf_blocks(raw[:len(raw) - 7]))
    env = dict(os.environ, **ENV)
    r = subprocess.run(f"{BIN} {' '.join(FILT)} -bu {cut} > /dev/null", shell=True, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Fatal Error: Truncated BAM record" in r.stderr, r.stderr[-300:]
    r = subprocess.run(f"{BIN} profile --label S -o {tmp_path / 'p.gz'} {cut}", shell=True, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Truncated BAM record" in r.stderr
    # a record length of 5 two thirds into the stream (found by walking the records from the header's end)
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    target = 2 * len(raw) // 3
    while p < target:
        p += 4 + struct.unpack_from("<i", raw, p)[0]
    bad = bytearray(raw)
    bad[p:p + 4] = struct.pack("<i", 5)
    badf = tmp_path / "bad.bam"
    badf.write_bytes(bgzf_blocks(bytes(bad)))
    for extra in ({}, {"MSX_HOST_UNPACK": "1"}):
        r = subprocess.run(f"{BIN} {' '.join(FILT)} -bu {badf} > /dev/null", shell=True, env=dict(env, **extra), stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"Fatal Error: Corrupt BAM record" in r.stderr, r.stderr[-300:]
