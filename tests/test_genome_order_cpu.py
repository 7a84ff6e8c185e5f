"""Feature order of `profile --genome` = key order of the reference's hash table
(msam_profile.c:779-805, zoeTools.c:202-363).  The golden vectors were produced by the
reference's own zoeTools.c (tests/golden/make_genome_order_vectors.py); the oracle restatement
and the host implementation (msamtools keyorder, a host-only hidden subcommand) must both
reproduce them.  Where oracle/_ref/libzoe_ref.so exists (the container with the reference
tree), the oracle is also checked against it on random name lists."""
import ctypes as C
import json
import os
import random
import subprocess

import pytest

import oracle_lib as orc
from conftest import GOLDEN, ROOT

VEC = json.load(open(os.path.join(GOLDEN, "genome_order_vectors.json")))["vectors"]
BIN = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools" + os.environ.get("MSX_BIN_SUFFIX", ""))
DEV = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev" + os.environ.get("MSX_BIN_SUFFIX", ""))       # generator and I/O self-tests (msh_dev.c)


def oracle_order(names):
    arr = (C.c_char_p * len(names))(*names)
    order = (C.c_int32 * max(len(names), 1))()
    lib = orc.lib()
    lib.orc_key_order.restype = C.c_int32
    n = lib.orc_key_order(arr, C.c_int32(len(names)), order)
    return [names[order[i]] for i in range(n)]


@pytest.mark.parametrize("v", VEC, ids=[v["name"] for v in VEC])
def test_oracle_key_order_matches_reference_vectors(v):
    names = [s.encode("latin-1") for s in v["insert"]]
    assert oracle_order(names) == [s.encode("latin-1") for s in v["keys"]]


@pytest.mark.parametrize("v", VEC, ids=[v["name"] for v in VEC])
def test_host_key_order_matches_reference_vectors(v):
    if not os.path.exists(BIN):
        pytest.skip("host binary not built")
    inp = b"".join(s.encode("latin-1") + b"\n" for s in v["insert"])
    out = subprocess.run([DEV, "keyorder"], input=inp, capture_output=True, check=True).stdout
    assert out.split(b"\n")[:-1] == [s.encode("latin-1") for s in v["keys"]]


def test_oracle_key_order_against_reference_library_random():
    so = os.path.join(ROOT, "oracle", "_ref", "libzoe_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (no reference tree here)")
    ref = C.CDLL(so)

    class TVec(C.Structure):
        _fields_ = [("elem", C.POINTER(C.c_char_p)), ("size", C.c_int), ("limit", C.c_int), ("last", C.c_char_p)]

    ref.zoeNewHash.restype = C.c_void_p
    ref.zoeSetHash.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    ref.zoeKeysOfHash.restype = C.POINTER(TVec)
    ref.zoeKeysOfHash.argtypes = [C.c_void_p]
    rnd = random.Random(7)
    one = C.c_int(1)
    for trial in range(40):
        n = rnd.choice([0, 1, 7, 8, 9, 31, 32, 33, 127, 128, 129, 500, 3000])
        pool = [bytes(rnd.randrange(33, 127) for _ in range(rnd.randrange(1, 30))) for _ in range(max(1, n // 2 + 1))]
        names = [rnd.choice(pool) for _ in range(n)]
        h = ref.zoeNewHash()
        for s in names:
            ref.zoeSetHash(h, s, C.addressof(one))
        tv = ref.zoeKeysOfHash(h).contents
        want = [tv.elem[i] for i in range(tv.size)]
        assert oracle_order(names) == want, (trial, n)
