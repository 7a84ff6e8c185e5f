"""The flat MD walk of k_aln_stats_flat (msx_md.h) equals the byte-at-a-time restatement of
mBamVector.c:112-118 (md_byte) on random byte strings."""
import os
import subprocess

from conftest import ROOT


def test_md_flat_walk_equals_md_byte(tmp_path):
    """The walk (carry-chain form of the token rule, 16 bytes per lane, ballot carries between lanes,
    prefix differences per string, several passes for long strings) simulated lane by lane on the host."""
    exe = str(tmp_path / "md_flat_test")
    # the header uses C++ references, so build the harness as C++
    subprocess.check_call(["g++", "-O2", "-x", "c++", "-o", exe, os.path.join(ROOT, "tests", "c", "md_flat_test.c")])
    out = subprocess.check_output([exe, "6000"]).decode()
    assert "bad=0" in out, out
