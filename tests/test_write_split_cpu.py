"""The parallel pwrite of msh_write_framed (MSX_WRITE_THREADS > 1) splits the framed blocks into per-thread ranges that
cover them exactly (msh_split.h), and the writer built on it leaves no hole in the file."""
import os
import subprocess

from conftest import ROOT


def test_split_ranges_cover_every_byte(tmp_path):
    exe = str(tmp_path / "split_test")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-o", exe, os.path.join(ROOT, "tests", "c", "split_test.c")])
    out = subprocess.check_output([exe]).decode()
    assert "bad=0" in out, out
