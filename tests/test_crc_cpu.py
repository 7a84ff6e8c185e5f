"""msh_crc32 -- the CRC-32 every BGZF payload read or written passes through -- equals zlib's crc32."""
import os
import subprocess

from conftest import ROOT


def test_crc32_by_carryless_multiplication_equals_zlib(tmp_path):
    host = os.path.join(ROOT, "msamtools_amd", "csrc", "host")
    exe = str(tmp_path / "crc_test")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", host, "-o", exe, os.path.join(ROOT, "tests", "c", "crc_test.c"),
                           os.path.join(host, "msh_io.c"), os.path.join(host, "msh_genome.c"), os.path.join(host, "msh_inflate.c"), "-lz", "-lpthread", "-lm"])
    for env in ({}, {"MSX_NO_PCLMUL": "1"}):
        out = subprocess.check_output([exe], env=dict(os.environ, **env)).decode()
        assert "bad=0" in out, out
