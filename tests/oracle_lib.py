"""ctypes binding of the CPU oracle (oracle/libmsx_oracle.so) for the tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this module: the oracle is the checker, never the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libmsx_oracle.so")

MULTI = {"all": 1, "equal": 2, "proportional": 3, "ignore": 4}
UNIT = {"rel": 1, "fpkm": 2, "tpm": 3, "ab": 4}


class OrcRecords(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("qname_off", C.c_void_p), ("qname", C.c_void_p), ("name_id", C.c_void_p),
        ("flag", C.c_void_p), ("rflags", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p),
        ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
        ("md_off", C.c_void_p), ("md", C.c_void_p),
        ("nm", C.c_void_p), ("as_", C.c_void_p),
    ]


class OrcFilterParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("min_length", "ppt", "max_clip", "rescore", "invert", "keep_unmapped", "besthit", "uniqhit")]


class OrcProfileStats(C.Structure):
    _fields_ = [
        ("insert_count", C.c_uint32), ("uniq_mapper_count", C.c_uint32),
        ("multi_mapper_count", C.c_uint32), ("purged_insert_count", C.c_uint32),
        ("iterations", C.c_int32), ("converged", C.c_int32), ("last_delta", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_ROOT, "oracle", "msx_oracle.c")
        if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.orc_filter.restype = C.c_int
        _lib.orc_profile.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_records(rec, name_id=None):
    """rec: any object with the samio.Records attributes (numpy arrays).
    Returns (OrcRecords, keepalive)."""
    r = OrcRecords()
    r.n = int(rec.flag.shape[0])
    keep = []

    def put(field, arr, dtype):
        if arr is None:
            setattr(r, field, None)
            return
        a = np.ascontiguousarray(arr, dtype=dtype)
        keep.append(a)
        setattr(r, field, _p(a))

    if name_id is None and getattr(rec, "qname_off", None) is not None:
        put("qname_off", rec.qname_off, np.uint32)
        put("qname", rec.qname if rec.qname.size else np.zeros(1, np.uint8), np.uint8)
        r.name_id = None
    else:
        r.qname_off = None
        r.qname = None
        put("name_id", name_id if name_id is not None else rec.name_id, np.int32)
    put("flag", rec.flag, np.uint16)
    put("rflags", rec.rflags, np.uint8)
    put("tid", rec.tid, np.int32)
    put("pos", rec.pos, np.int32)
    put("cigar_off", rec.cigar_off, np.uint32)
    put("cigar", rec.cigar if rec.cigar.size else np.zeros(1, np.uint32), np.uint32)
    put("md_off", rec.md_off, np.uint32)
    put("md", rec.md if rec.md.size else np.zeros(1, np.uint8), np.uint8)
    put("nm", rec.nm, np.int32)
    put("as_", rec.as_, np.int32)
    return r, keep


def filter_params(l=0, p=None, ppt=None, z=None, rescore=False, invert=False,
                  keep_unmapped=False, besthit=False, uniqhit=False):
    """CLI options -> thresholds exactly as msam_filter.c:420-457 derives them."""
    fp = OrcFilterParams()
    fp.min_length = int(l or 0)
    fp.ppt = 10 * int(p) if p is not None else (int(ppt) if ppt is not None else 0)
    fp.max_clip = 100 - int(z) if z is not None else 100
    fp.rescore, fp.invert, fp.keep_unmapped = int(rescore), int(invert), int(keep_unmapped)
    fp.besthit, fp.uniqhit = int(besthit), int(uniqhit)
    return fp


def aln_stats(rec, name_id=None):
    r, keep = make_records(rec, name_id)
    n = r.n
    out = {k: np.zeros(n, np.int32) for k in ("length", "qlen", "qclip", "edit")}
    status = np.zeros(n, np.uint8)
    lib().orc_aln_stats(C.byref(r), _p(out["length"]), _p(out["qlen"]), _p(out["qclip"]),
                        _p(out["edit"]), _p(status))
    out["status"] = status
    return out


def run_filter(rec, name_id=None, **opts):
    """Returns dict(rc, emit, as_out, err_record)."""
    r, keep = make_records(rec, name_id)
    fp = filter_params(**opts)
    emit = np.zeros(max(r.n, 1), np.int32)
    as_out = np.zeros(max(r.n, 1), np.int32)
    n_emit = C.c_int64(0)
    err = C.c_int64(-1)
    rc = lib().orc_filter(C.byref(r), C.byref(fp), _p(emit), C.byref(n_emit), _p(as_out), C.byref(err))
    return dict(rc=rc, emit=emit[:n_emit.value].copy(), as_out=as_out[:r.n].copy(), err_record=err.value)


def run_profile(rec, n_features, multi="proportional", sel=None, fmap=None, name_id=None):
    """Returns dict(abundance, ui, stats)."""
    r, keep = make_records(rec, name_id)
    ab = np.zeros(max(n_features, 1), np.float64)
    ui = np.zeros(max(n_features, 1), np.uint32)
    st = OrcProfileStats()
    selp, nsel = None, 0
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        selp, nsel = _p(sel if sel.size else np.zeros(1, np.int32)), int(sel.size)
    fm = None if fmap is None else np.ascontiguousarray(fmap, dtype=np.int32)
    lib().orc_profile(C.byref(r), selp, C.c_int64(nsel), _p(fm), C.c_int32(n_features),
                      C.c_int32(MULTI[multi]), _p(ab), _p(ui), C.byref(st))
    return dict(abundance=ab[:n_features].copy(), ui=ui[:n_features].copy(), stats=st)


def profile_finish(abundance, feature_len, stats, unit="rel", nolen=False, total=-1,
                   mincount=-1, multi="proportional"):
    """Returns (values[Unknown, features...], purged_inserts, effective_inserts)."""
    nf = len(abundance)
    vals = np.zeros(nf + 1, np.float64)
    vals[1:] = abundance
    fl = np.ascontiguousarray(feature_len, dtype=np.uint32)
    length_normalize = 1
    if unit in ("rel", "ab"):
        length_normalize = 0 if nolen else 1     # msam_profile.c:752-755
    purged = C.c_double(0)
    eff = C.c_double(0)
    lib().orc_profile_finish(_p(vals), C.c_int32(nf), _p(fl), C.c_int32(UNIT[unit]),
                             C.c_int32(length_normalize), C.c_int32(total), C.c_int32(mincount),
                             C.c_int32(MULTI[multi]), C.byref(stats), C.byref(purged), C.byref(eff))
    return vals, purged.value, eff.value


def coverage(rec, target_len):
    r, keep = make_records(rec)
    off = np.zeros(len(target_len) + 1, np.int64)
    off[1:] = np.cumsum(np.asarray(target_len, dtype=np.int64))
    cov = np.zeros(max(int(off[-1]), 1), np.int32)
    lib().orc_coverage(C.byref(r), _p(off), C.c_int32(len(target_len)), _p(cov))
    return [cov[off[i]:off[i + 1]].copy() for i in range(len(target_len))]


def make_records_slice(rec, name_id, lo, hi):
    """View of records [lo, hi) sharing the payload arrays (offsets stay absolute)."""
    class _V:
        pass
    v = _V()
    v.qname_off = None
    v.qname = None
    v.name_id = np.ascontiguousarray(name_id[lo:hi])
    v.flag = rec.flag[lo:hi]
    v.rflags = rec.rflags[lo:hi]
    v.tid = rec.tid[lo:hi]
    v.pos = rec.pos[lo:hi]
    v.cigar_off = rec.cigar_off[lo:hi + 1]
    v.cigar = rec.cigar
    v.md_off = rec.md_off[lo:hi + 1]
    v.md = rec.md
    v.nm = rec.nm[lo:hi]
    v.as_ = rec.as_[lo:hi]
    return v


def summary_table(rec, target_len, edge=0):
    """msamtools summary's per-alignment table (msam_summary.c:42-74): (record numbers, [query_length, glocal_len, match,
    edit] per row)."""
    r, keep = make_records(rec)
    tl = np.ascontiguousarray(target_len, dtype=np.uint32)
    sel = np.zeros(max(r.n, 1), np.int64)
    vals = np.zeros(4 * max(r.n, 1), np.int32)
    f = lib().orc_summary_table
    f.restype = C.c_int64
    n = f(C.byref(r), _p(tl), C.c_uint32(edge), _p(sel), _p(vals))
    return sel[:n].copy(), vals[:4 * n].reshape(n, 4).copy()


def summary_lines(rec, names, target_names, target_len, edge=0):
    """The table as the text the reference prints (msam_summary.c:71)."""
    sel, v = summary_table(rec, target_len, edge)
    out = []
    for k, i in enumerate(sel):
        ql, gl, m, e = (int(x) for x in v[k])
        with np.errstate(divide="ignore", invalid="ignore"):
            ident = float(np.float64(100.0) - np.float64(100.0) * np.float64(e) / np.float64(gl))
        out.append("%s\t%d\t%s\t%d\t%d\t%.1f" % (names[i], ql, target_names[rec.tid[i]], gl, m, ident))
    return out


def summary_stats(rec, target_len, which, edge=0):
    """--stats {mapped|unmapped|edit|score}: the distribution as the reference prints it (msam_summary.c:76-135)."""
    r, keep = make_records(rec)
    tl = np.ascontiguousarray(target_len, dtype=np.uint32)
    dist = np.zeros(4097, np.int64)
    lib().orc_summary_stats(C.byref(r), _p(tl), C.c_uint32(edge), C.c_int(("mapped", "unmapped", "edit", "score").index(which)), _p(dist))
    out = ["%d\t%d" % (i, dist[i]) for i in range(4096) if dist[i] > 0]
    if dist[4096] > 0:
        out.append("4096+\t%d" % dist[4096])
    return out


def summary_count(rec, name_id=None):
    """--count: QNAME groups among the mapped records (msam_summary.c:19-40)."""
    r, keep = make_records(rec, name_id)
    f = lib().orc_summary_count
    f.restype = C.c_int64
    return int(f(C.byref(r)))
