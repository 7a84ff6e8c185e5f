#!/usr/bin/env python3
"""Generates tests/golden/libdeflate_blocks.bin + .json: raw DEFLATE streams written by libdeflate (the encoder htslib
uses when built with it, and samtools' default in most distributions) at levels 1, 6 and 12, for the device inflater's
tests (tests/test_gpu_inflate.py).  Every multi-batch input of the other tests is written by this repository's own
writers (zlib, or the device encoder); libdeflate chooses block splits, code lengths and match shapes zlib never does.
Run in the build container (libdeflate.so.0 is there; the GPU box only reads the committed data):
    python tests/golden/make_libdeflate_blocks.py
Data only: inputs are regenerated from the seed below, the file holds the compressed bytes."""
import ctypes
import json
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs():
    """name -> bytes (<= 65280 each); must stay in step with tests/test_gpu_inflate.py::libdeflate_inputs"""
    rng = np.random.default_rng(20261003)
    out = {}

    def bam_like(n_bytes, seq):
        buf = bytearray()
        k = 0
        while len(buf) < n_bytes:
            name = b"read%07d" % k
            for h in range(int(rng.integers(1, 7))):
                core = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
                buf += (60 + len(name)).to_bytes(4, "little") + core + name + b"\0" + bytes([100 << 4 & 255, 6, 0, 0])
                if seq:
                    buf += rng.integers(0, 256, 50, dtype=np.uint8).tobytes() + bytes(rng.integers(33, 74, 100, dtype=np.uint8))
                buf += b"NMC" + bytes([int(rng.integers(0, 4))]) + b"ASC" + bytes([int(rng.integers(90, 101))]) + b"MDZ100\0"
            k += 1
        return bytes(buf[:n_bytes])
    out["bam lean"] = bam_like(65280, False)
    out["bam with seq and qual"] = bam_like(65280, True)
    out["short"] = bam_like(777, False)
    out["text"] = b"".join(b"@SQ\tSN:contig_%06d\tLN:%d\n" % (i, 1000 + 7 * i) for i in range(2300))[:65280]
    out["random"] = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    out["zeros"] = bytes(65280)
    out["skewed"] = bytes(np.minimum(rng.geometric(0.3, 65280), 255).astype(np.uint8))
    return out


def main():
    L = ctypes.CDLL("libdeflate.so.0")
    L.libdeflate_alloc_compressor.restype = ctypes.c_void_p
    L.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
    L.libdeflate_deflate_compress.restype = ctypes.c_size_t
    L.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.libdeflate_free_compressor.argtypes = [ctypes.c_void_p]
    blob = bytearray()
    index = []
    for level in (1, 6, 12):
        c = L.libdeflate_alloc_compressor(level)
        for name, data in inputs().items():
            cap = len(data) + 1024
            buf = ctypes.create_string_buffer(cap)
            n = L.libdeflate_deflate_compress(c, data, len(data), buf, cap)
            assert n > 0
            comp = buf.raw[:n]
            assert zlib.decompress(comp, -15) == data
            index.append({"name": name, "level": level, "offset": len(blob), "length": n, "isize": len(data), "crc32": zlib.crc32(data)})
            blob += comp
        L.libdeflate_free_compressor(c)
    open(os.path.join(HERE, "libdeflate_blocks.bin"), "wb").write(bytes(blob))
    json.dump({"_comment": "raw DEFLATE streams from libdeflate.so.0 (make_libdeflate_blocks.py); inputs are regenerated from its seed",
               "blocks": index}, open(os.path.join(HERE, "libdeflate_blocks.json"), "w"), indent=1)
    print(len(index), "streams,", len(blob), "bytes")


if __name__ == "__main__":
    main()
