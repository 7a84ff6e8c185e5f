"""The host pipeline's threads under ThreadSanitizer (no GPU): the decode stage (BGZF inflate on the pool, the speculative
record chase, SAM text parsed in chunks on all threads), the slot queues and the stream writer, through
`msamtools-dev-tsan pipetest / digest / restream` (make tsan).  Round 5: the SAM-text path shared one static cache word
between the parsing threads (msh_hdr_name2tid) -- found by this run."""
import os
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "msamtools_amd", "csrc", "host")
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
TSAN = DEV + "-tsan"


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    r = subprocess.run(["make", "-C", HOST, "all", "tsan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0 or not os.path.exists(TSAN):
        pytest.skip("no ThreadSanitizer build here: " + r.stdout.decode()[-300:])
    d = tmp_path_factory.mktemp("tsan")
    out = {}
    for name, extra in (("b", ["-b"]), ("u", ["-u", "--seq"])):
        out[name] = str(d / f"in_{name}.bam")
        with open(out[name], "wb") as fh:
            subprocess.check_call([DEV, "synth", "--groups", "30000", "--refs", "400"] + extra, stdout=fh)
    out["sam"] = str(d / "in.sam")
    with open(out["sam"], "wb") as fh:
        subprocess.check_call([DEV, "recode", "-h", out["b"]], stdout=fh)
    probe = subprocess.run([TSAN, "digest", out["b"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if b"unexpected memory mapping" in probe.stderr:
        pytest.skip("ThreadSanitizer does not run under this kernel's address-space layout")
    return out


def tsan(args, **env):
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0", **{k: str(v) for k, v in env.items()})
    r = subprocess.run([TSAN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert b"ThreadSanitizer" not in r.stderr, r.stderr.decode()[:3000]
    assert r.returncode == 0, r.stderr.decode()[-500:]
    return r.stdout.decode()


@pytest.mark.parametrize("which,mode,env", [
    ("b", "1 1", dict(MSX_THREADS=8, MSX_BATCH_BYTES=600_000)),
    ("u", "0 1", dict(MSX_THREADS=5, MSX_BATCH_BYTES=300_000, MSX_BATCH_RECORDS=20_000)),
    ("u", "2 0", dict(MSX_THREADS=16, MSX_BATCH_RECORDS=5_000)),
    ("sam", "1 1", dict(MSX_THREADS=8, MSX_SAM_CHUNK=100_000)),
    ("sam", "0 1", dict(MSX_THREADS=3, MSX_SAM_CHUNK=50_000, MSX_BATCH_BYTES=500_000)),
])
def test_decode_stage_has_no_data_race(files, which, mode, env):
    out = tsan(["pipetest"] + mode.split() + [files[which]], **env).split("\n")
    assert out[0].split()[1:5] == out[1].split()[1:5]          # pipeline == serial reader: records, pools, hashes


def test_digest_and_stream_writer_have_no_data_race(files):
    want = subprocess.check_output([DEV, "digest", "--full", files["u"]]).decode()
    assert tsan(["digest", "--full", files["u"]], MSX_THREADS=8, MSX_BATCH_RECORDS=20_000) == want
    r = subprocess.run(f"MSX_THREADS=8 MSX_BATCH_RECORDS=20000 {TSAN} restream -b {files['u']} | {DEV} digest --full /dev/stdin", shell=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert b"ThreadSanitizer" not in r.stderr and r.stdout.decode() == want
