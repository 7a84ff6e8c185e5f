"""The record-stream digest of `msamtools digest` (msh_dev.c: digest_main), computed with numpy from SoA fields.

digest = sum over the records i = 0.. of (i + 1) * g(record i) mod 2^64,
g = mix64(flag + u32(tid) * K1 + u32(pos) * K2) xor FNV-1a(QNAME).  Order-sensitive (the weights), so "the same
records in the same order" for tens of millions of records is one comparison of two 64-bit numbers.
Test infrastructure: the expected side of the command-line parity tests and of bench.py's e2e block.
"""
import numpy as np

_K1 = np.uint64(0x9e3779b97f4a7c15)
_K2 = np.uint64(0xc2b2ae3d27d4eb4f)
_M1 = np.uint64(0xff51afd7ed558ccd)
_M2 = np.uint64(0xc4ceb9fe1a85ec53)
_FNV_OFF = np.uint64(1469598103934665603)
_FNV_PRIME = np.uint64(1099511628211)
_S33 = np.uint64(33)


def _mix(x):
    x = x ^ (x >> _S33)
    x = x * _M1
    x = x ^ (x >> _S33)
    x = x * _M2
    return x ^ (x >> _S33)


def fnv_sim_names(group_index):
    """FNV-1a of the QNAMEs `sim%08d` (msamtools synth, msh_dev.c: synth_worker) of the given group indices."""
    g = np.asarray(group_index, dtype=np.int64)
    out = np.empty(g.shape, dtype=np.uint64)
    width = np.full(g.shape, 8, dtype=np.int64)                 # %08d: at least eight digits
    for extra in range(9, 19):
        width[g >= 10 ** (extra - 1)] = extra
    with np.errstate(over="ignore"):
        for w in np.unique(width):
            m = width == w
            v = g[m]
            h = np.full(v.shape, _FNV_OFF, dtype=np.uint64)
            for ch in b"sim":
                h = (h ^ np.uint64(ch)) * _FNV_PRIME
            for d in range(int(w) - 1, -1, -1):
                digit = (v // 10 ** d) % 10
                h = (h ^ (digit.astype(np.uint64) + np.uint64(48))) * _FNV_PRIME
            out[m] = h
    return out


def fnv_names(names):
    """FNV-1a of arbitrary byte-string names (a Python loop: small inputs)."""
    out = np.empty(len(names), dtype=np.uint64)
    for i, nm in enumerate(names):
        h = 1469598103934665603
        for ch in (nm if isinstance(nm, bytes) else nm.encode()):
            h = ((h ^ ch) * 1099511628211) & 0xffffffffffffffff
        out[i] = h
    return out


def stream_digest(flag, tid, pos, name_hash, sel=None):
    """Digest of the records sel[0], sel[1], ... (all records in order when sel is None)."""
    if sel is not None:
        sel = np.asarray(sel, dtype=np.int64)
        flag, tid, pos, name_hash = flag[sel], tid[sel], pos[sel], name_hash[sel]
    with np.errstate(over="ignore"):
        v = flag.astype(np.uint64) + tid.astype(np.uint32).astype(np.uint64) * _K1 + \
            pos.astype(np.uint32).astype(np.uint64) * _K2
        g = _mix(v) ^ name_hash.astype(np.uint64)
        w = np.arange(1, g.size + 1, dtype=np.uint64)
        return int(np.sum(w * g, dtype=np.uint64)), int(g.size)


def synth_digest(hs, sel=None, first_group=0):
    """hs: msamtools_amd.HostSynth (or anything with flag/tid/pos/group_off): the digest `msamtools digest`
    prints for `msamtools synth` of the same parameters, restricted to the records `sel`."""
    counts = np.diff(hs.group_off.astype(np.int64))
    gidx = np.repeat(np.arange(counts.size, dtype=np.int64) + first_group, counts)
    return stream_digest(hs.flag, hs.tid, hs.pos, fnv_sim_names(gidx), sel)
