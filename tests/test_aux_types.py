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
