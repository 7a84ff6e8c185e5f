"""The SWAR (dword-at-a-time) MD walk used by k_aln_stats_filter equals the
byte-at-a-time restatement of mBamVector.c:112-118 on random byte strings."""
import os
import subprocess

from conftest import ROOT


def test_md_word_equals_md_byte(tmp_path):
    exe = str(tmp_path / "md_swar_test")
    # the header uses C++ references, so build the harness as C++
    subprocess.check_call(["g++", "-O2", "-x", "c++", "-o", exe, os.path.join(ROOT, "tests", "c", "md_swar_test.c")])
    out = subprocess.check_output([exe, "3000000"]).decode()
    assert "bad=0" in out, out


def test_md_flat_walk_equals_md_byte(tmp_path):
    """The flat walk of k_aln_stats_flat (carry-chain form of the token rule, 16 bytes per lane,
    ballot carries between lanes, prefix differences per string, several passes for long strings)
    simulated lane by lane on the host against the byte-at-a-time rule."""
    exe = str(tmp_path / "md_flat_test")
    subprocess.check_call(["g++", "-O2", "-x", "c++", "-o", exe, os.path.join(ROOT, "tests", "c", "md_flat_test.c")])
    out = subprocess.check_output([exe, "6000"]).decode()
    assert "bad=0" in out, out
