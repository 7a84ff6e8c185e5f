#!/usr/bin/env python3
"""Generates tests/golden/genome_order_vectors.json: for several sets of genome names, the order
in which the REFERENCE's hash table (zoeTools.c, compiled in place into oracle/_ref/libzoe_ref.so
by `make -C oracle ref`) returns its keys after the names were inserted one by one -- i.e. the
feature order of `msamtools profile --genome` (msam_profile.c:771-852).  Runs only where the
reference tree is mounted; the JSON it writes is data (inputs and expected outputs)."""
import ctypes as C
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzoe_ref.so"))


class TVec(C.Structure):
    _fields_ = [("elem", C.POINTER(C.c_char_p)), ("size", C.c_int), ("limit", C.c_int), ("last", C.c_char_p)]


lib.zoeNewHash.restype = C.c_void_p
lib.zoeSetHash.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
lib.zoeKeysOfHash.restype = C.POINTER(TVec)
lib.zoeKeysOfHash.argtypes = [C.c_void_p]


def ref_order(names):
    h = lib.zoeNewHash()
    one = C.c_int(1)
    for n in names:
        lib.zoeSetHash(h, n, C.addressof(one))
    v = lib.zoeKeysOfHash(h).contents
    return [v.elem[i] for i in range(v.size)]


def cases():
    rnd = random.Random(20261003)
    out = []
    out.append(("five", [b"genomeA", b"genomeB", b"genomeC", b"genomeD", b"genomeE"]))
    out.append(("seven_then_repeat", [b"g%d" % i for i in range(7)] + [b"g3", b"g0"]))
    out.append(("eight_triggers_first_growth", [b"taxon_%02d" % i for i in range(8)]))
    out.append(("nine", [b"taxon_%02d" % i for i in range(9)]))
    out.append(("thirtythree", [b"GCF_%09d.1" % rnd.randrange(10**9) for _ in range(33)]))
    out.append(("strains_130", [b"Escherichia_coli_str_%d" % i if i % 3 else b"B.longum.%d" % i for i in range(130)]))
    alpha = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_.-|"
    out.append(("random_600", list(dict.fromkeys(
        bytes(rnd.choice(alpha) for _ in range(rnd.randrange(1, 40))) for _ in range(600)))))
    out.append(("high_bytes", [bytes([0x41 + i % 20, 0xC3, 0xA9 - i % 7, 0x30 + i % 10]) + b"_%d" % i for i in range(40)]))
    out.append(("with_duplicates_2000", [b"MAG_%04d" % rnd.randrange(700) for _ in range(2000)]))
    return out


vectors = []
for name, names in cases():
    order = ref_order(names)
    vectors.append({"name": name, "insert": [n.decode("latin-1") for n in names],
                    "keys": [n.decode("latin-1") for n in order]})
json.dump({"_source": "reference zoeTools.c (zoeNewHash / zoeSetHash / zoeKeysOfHash), compiled in place; strings are "
                      "latin-1 decodings of the raw bytes", "vectors": vectors},
          open(os.path.join(ROOT, "tests", "golden", "genome_order_vectors.json"), "w"), indent=0)
print("wrote", len(vectors), "vectors:", [(v["name"], len(v["keys"])) for v in vectors])
