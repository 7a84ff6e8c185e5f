"""msh_fmt_g8 -- the profile text's "%.8g" without printf (msamtools_amd/csrc/host/msh_fmt.h; mMatrix.c:359-376 is the line it
writes) -- produces snprintf's bytes: random bit patterns, the magnitudes a profile holds, decimal values next to rounding ties,
powers of ten and their neighbours, exact ties, zeros, subnormals, infinities, NaN."""
import os
import subprocess

from conftest import ROOT


def test_g8_formatter_equals_printf(tmp_path):
    exe = str(tmp_path / "fmt_g8_test")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "c", "fmt_g8_test.c"), "-lm"])
    out = subprocess.check_output([exe, "1200000"]).decode()
    assert "bad=0" in out, out
