"""msh_fast_inflate -- the DEFLATE decoder BGZF blocks go through before zlib is asked -- equals zlib's inflate
on valid streams and survives invalid ones (AddressSanitizer + UBSan build)."""
import os
import subprocess

from conftest import ROOT


def test_fast_inflate_equals_zlib_and_survives_corruption(tmp_path):
    host = os.path.join(ROOT, "msamtools_amd", "csrc", "host")
    exe = str(tmp_path / "inflate_test")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu99", "-I", host,
                           "-o", exe, os.path.join(ROOT, "tests", "c", "inflate_test.c"), os.path.join(host, "msh_inflate.c"), "-lz"])
    r = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "rejected=0 bad=0" in out, out[-2000:]
