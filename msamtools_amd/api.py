"""Host-side Python mirror of the filter -> profile seam, over the C ABI.

Used by the tests and bench.py; the production host is the C command line in
msamtools_amd/csrc/host (same ABI).  Naming follows the reference: a *pool* is
the set of alignments of one QNAME (mBamPool), `filter_opts` are the CLI
switches of `msamtools filter` (msam_filter.c:304-347), `multi` the --multi
mode of `msamtools profile` (msam_profile.c:712-728).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def filter_params(l=0, p=None, ppt=None, z=None, rescore=False, invert=False,
                  keep_unmapped=False, besthit=False, uniqhit=False, fatal_pool_partial=False):
    """CLI options -> msx_filter_params, validated as msam_filter.c:398-457 does."""
    if invert and (besthit or uniqhit):
        raise ValueError("--invert cannot be combined with --besthit or --uniqhit")
    if besthit and uniqhit:
        raise ValueError("--besthit cannot be combined with --uniqhit")
    if p is not None and ppt is not None:
        raise ValueError("-p cannot be combined with --ppt")
    fp = L.FilterParams()
    fp.ppt = 0
    if p is not None:
        if not 0 <= int(p) <= 100:
            raise ValueError("-p must be in the range [0,100]")
        fp.ppt = 10 * int(p)
    elif ppt is not None:
        if not -1000 <= int(ppt) <= 1000:
            raise ValueError("--ppt must be in the range [-1000,1000]")
        fp.ppt = int(ppt)
    fp.max_clip = 100
    if z is not None:
        fp.max_clip = 100 - int(z)
        if not 0 <= fp.max_clip <= 100:
            raise ValueError("-z must be in the range [0,100]")
    fp.min_length = int(l or 0)
    if fp.min_length < 0:
        raise ValueError("-l must be a non-negative integer")
    fp.rescore, fp.invert, fp.keep_unmapped = int(bool(rescore)), int(bool(invert)), int(bool(keep_unmapped))
    fp.besthit, fp.uniqhit = int(bool(besthit)), int(bool(uniqhit))
    fp.fatal_pool_partial = int(bool(fatal_pool_partial))
    return fp


def dist_unique_id():
    """ncclGetUniqueId on the calling rank (rank 0 distributes the 128 bytes)."""
    lib = L.load()
    buf = (C.c_uint8 * L.DIST_ID_BYTES)()
    L.check(None, lib.msx_dist_unique_id(buf))
    return bytes(buf)


class Context:
    """One GPU (msx_ctx).  Raises if there is no gfx950 device."""

    def __init__(self, device_id=0):
        self.lib = L.load()
        h = C.c_void_p()
        L.check(None, self.lib.msx_ctx_create(C.byref(h), int(device_id)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.msx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        return L.check(self.h, rc)

    @property
    def stream(self):
        return self.lib.msx_ctx_stream(self.h)

    def sync(self):
        self.check(self.lib.msx_ctx_sync(self.h))

    # several GPUs: one rank per process (msx_dist_*) ------------------------
    def dist_init(self, id_bytes, rank, world):
        buf = (C.c_uint8 * L.DIST_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        self.check(self.lib.msx_dist_init(self.h, buf, int(rank), int(world)))

    def dist_init_env(self):
        self.check(self.lib.msx_dist_init_env(self.h))

    @property
    def rank(self):
        return self.lib.msx_dist_rank(self.h)

    @property
    def world(self):
        return self.lib.msx_dist_world(self.h)

    def barrier(self):
        """Stream sync + (with a communicator) a one-element all-reduce over the ranks."""
        self.check(self.lib.msx_dist_barrier(self.h))

    def max_over_ranks(self, value):
        v = C.c_double(float(value))
        self.check(self.lib.msx_dist_max_f64(self.h, C.byref(v)))
        return v.value

    def sum_over_ranks(self, value):
        v = C.c_int64(int(value))
        self.check(self.lib.msx_dist_sum_i64(self.h, C.byref(v)))
        return v.value

    # raw device memory -----------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.msx_dev_alloc(self.h, C.byref(p), int(nbytes)))
        return p.value

    def free(self, ptr):
        if ptr:
            self.lib.msx_dev_free(self.h, C.c_void_p(ptr))

    def zero(self, ptr, nbytes):
        self.check(self.lib.msx_dev_zero(self.h, C.c_void_p(ptr), int(nbytes)))

    def to_host(self, ptr, count, dtype):
        out = np.empty(int(count), dtype=dtype)
        if out.nbytes:
            self.check(self.lib.msx_dev_to_host(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def to_dev(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            self.check(self.lib.msx_host_to_dev(self.h, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    # timing ---------------------------------------------------------------
    def timing(self, on=True):
        self.check(self.lib.msx_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self.check(self.lib.msx_timing_reset(self.h))

    def timing_get(self, name):
        ms = C.c_double(0)
        cnt = C.c_int64(0)
        self.check(self.lib.msx_timing_get(self.h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def timing_bytes(self, name):
        b = C.c_int64(0)
        self.check(self.lib.msx_timing_get_bytes(self.h, name.encode(), C.byref(b)))
        return b.value


_FIELDS = (("flag", np.uint16), ("rflags", np.uint8), ("tid", np.int32), ("pos", np.int32),
           ("cigar_off", np.uint32), ("cigar", np.uint32), ("md_off", np.uint32), ("md", np.uint8),
           ("nm", np.int32), ("as_", np.int32), ("group_off", np.uint32))


def host_batch_struct(rec, group_off=None, filter_pools=False):
    """numpy SoA (attributes as in tests/samio.Records) -> (msx_batch with host pointers, keepalive).
    filter_pools: group_off follow the rule of msam_filter.c:120-125,170 (msx_batch.pool_rule)."""
    b = L.Batch()
    b.pool_rule = L.POOLS_FILTER if filter_pools else L.POOLS_PROFILE
    keep = []
    b.n_records = int(rec.flag.shape[0])
    for name, dt in _FIELDS:
        src = group_off if name == "group_off" else getattr(rec, name, None)
        if src is None:
            setattr(b, name, None)
            continue
        a = np.ascontiguousarray(src, dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        keep.append(a)
        setattr(b, name, a.ctypes.data_as(C.c_void_p))
    b.n_groups = 0 if group_off is None else int(len(group_off) - 1)
    b.qname_hash = None
    return b, keep


class DeviceBatch:
    """A record batch resident in HBM (msx_batch with device pointers)."""

    def __init__(self, ctx, b, sizes=None):
        self.ctx = ctx
        self.b = b
        self.sizes = sizes

    @classmethod
    def upload(cls, ctx, rec, group_off=None, filter_pools=False):
        hb, keep = host_batch_struct(rec, group_off, filter_pools)
        db = L.Batch()
        ctx.check(ctx.lib.msx_batch_upload(ctx.h, C.byref(hb), C.byref(db)))
        return cls(ctx, db)

    @classmethod
    def synth(cls, ctx, seed, n_groups, n_refs, mean_extra_hits=4, first_group=0):
        sp = L.SynthParams(seed, n_groups, n_refs, mean_extra_hits, first_group)
        db = L.Batch()
        sz = L.SynthSizes()
        ctx.check(ctx.lib.msx_synth_device(ctx.h, C.byref(sp), C.byref(db), C.byref(sz)))
        return cls(ctx, db, sz)

    @property
    def n_records(self):
        return int(self.b.n_records)

    @property
    def n_groups(self):
        return int(self.b.n_groups)

    def fetch(self, name, count, dtype):
        return self.ctx.to_host(getattr(self.b, name), count, dtype)

    def to_host(self):
        """Copy the whole batch back as a dict of numpy arrays."""
        n, ng = self.n_records, self.n_groups
        out = {}
        for name, dt in _FIELDS:
            if not getattr(self.b, name):
                continue
            if name == "cigar":
                cnt = int(out["cigar_off"][n]) if n else 0
            elif name == "md":
                cnt = int(out["md_off"][n]) if n else 0
            elif name in ("cigar_off", "md_off"):
                cnt = n + 1
            elif name == "group_off":
                cnt = ng + 1
            else:
                cnt = n
            out[name] = self.fetch(name, cnt, dt)
        return out

    def free(self):
        if self.b is not None:
            self.ctx.lib.msx_batch_free(self.ctx.h, C.byref(self.b))
            self.b = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class HostSynth:
    """Host twin of the synthetic generator (msx_synth_host): numpy views."""

    def __init__(self, seed, n_groups, n_refs, mean_extra_hits=4, first_group=0):
        lib = L.load()
        sp = L.SynthParams(seed, n_groups, n_refs, mean_extra_hits, first_group)
        hb = L.Batch()
        sz = L.SynthSizes()
        L.check(None, lib.msx_synth_host(C.byref(sp), C.byref(hb), C.byref(sz)))
        n, ng = int(sz.n_records), int(n_groups)

        self._hb = hb      # numpy views below alias the C buffers; freed in __del__

        def grab(ptr, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            buf = (C.c_uint8 * (cnt * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt, count=cnt)
        self.flag = grab(hb.flag, n, np.uint16)
        self.rflags = grab(hb.rflags, n, np.uint8)
        self.tid = grab(hb.tid, n, np.int32)
        self.pos = grab(hb.pos, n, np.int32)
        self.nm = grab(hb.nm, n, np.int32)
        self.as_ = grab(hb.as_, n, np.int32)
        self.cigar_off = grab(hb.cigar_off, n + 1, np.uint32)
        self.md_off = grab(hb.md_off, n + 1, np.uint32)
        self.cigar = grab(hb.cigar, int(sz.n_cigar), np.uint32)
        self.md = grab(hb.md, int(sz.n_md), np.uint8)
        self.group_off = grab(hb.group_off, ng + 1, np.uint32)
        self.qname_off = None
        self.qname = None
        # the QNAME of a synthetic record is its group index
        self.name_id = np.repeat(np.arange(ng, dtype=np.int32) + np.int32(first_group % (1 << 31)),
                                 np.diff(self.group_off).astype(np.int64))
        self.n_records, self.n_groups = n, ng

    def __del__(self):
        try:
            hb = getattr(self, "_hb", None)
            if hb is not None:
                self._hb = None
                L.load().msx_synth_host_free(C.byref(hb))
        except Exception:
            pass


def bgzf_blocks(payloads, datas=None, crcs=None, gap=0):
    """(compressed buffer, BgzfBlock array, total inflated length) for raw DEFLATE payloads laid out back to back
    (`gap` bytes of filler between them); out_len / crc32 from `datas` (the expected bytes) unless given"""
    import zlib
    n = len(payloads)
    arr = (L.BgzfBlock * max(n, 1))()
    comp = bytearray()
    uo = 0
    for i, pl in enumerate(payloads):
        comp += b"\xa5" * gap
        arr[i].in_off = len(comp)
        arr[i].in_len = len(pl)
        arr[i].out_off = uo
        arr[i].out_len = len(datas[i])
        arr[i].crc32 = (crcs[i] if crcs is not None else zlib.crc32(datas[i])) & 0xffffffff
        comp += pl
        uo += len(datas[i])
    return bytes(comp), arr, uo


def bgzf_inflate(ctx, comp, blocks, n_blocks, total_out):
    """msx_bgzf_inflate on host data: (output bytes, status u32[n_blocks], blocks refused)"""
    comp = bytes(comp)
    d_comp = ctx.alloc(len(comp) + 16)
    d_blk = ctx.alloc(C.sizeof(L.BgzfBlock) * max(n_blocks, 1))
    d_out = ctx.alloc(total_out + 64)
    d_st = ctx.alloc(4 * max(n_blocks, 1))
    try:
        ctx.to_dev(d_comp, np.frombuffer(comp + b"\0" * 16, np.uint8))
        ctx.to_dev(d_blk, np.frombuffer(bytes(blocks), np.uint8))
        ctx.zero(d_out, total_out + 64)
        refused = C.c_int64(-1)
        ctx.check(ctx.lib.msx_bgzf_inflate(ctx.h, C.c_void_p(d_comp), len(comp), C.c_void_p(d_blk), n_blocks,
                                           C.c_void_p(d_out), C.c_void_p(d_st), C.byref(refused)))
        out = ctx.to_host(d_out, total_out + 64, np.uint8)
        st = ctx.to_host(d_st, max(n_blocks, 1), np.uint32)[:n_blocks]
        return out, st, refused.value
    finally:
        for p in (d_comp, d_blk, d_out, d_st):
            ctx.free(p)


def bgzf_deflate(ctx, data, level=0):
    """msx_bgzf_deflate on host data: (bytes of finished BGZF blocks, number of blocks)"""
    data = bytes(data)
    cap = int(ctx.lib.msx_bgzf_bound(len(data), level)) + 64
    d_in = ctx.alloc(len(data) + 64)
    d_out = ctx.alloc(cap)
    try:
        ctx.to_dev(d_in, np.frombuffer(data + b"\0" * 64, np.uint8))
        n_out, n_blk = C.c_int64(-1), C.c_int64(-1)
        ctx.check(ctx.lib.msx_bgzf_deflate(ctx.h, C.c_void_p(d_in), len(data), level, C.c_void_p(d_out), cap,
                                           C.byref(n_out), C.byref(n_blk)))
        out = ctx.to_host(d_out, max(n_out.value, 1), np.uint8)[:n_out.value].tobytes()
        return out, n_blk.value
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


def bgzf_split(stream):
    """the BGZF blocks of a byte string: [(payload bytes, isize, crc32)], checking every header on the way"""
    out, p = [], 0
    while p < len(stream):
        h = stream[p:p + 18]
        assert len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[10:12] == b"\x06\x00" and h[12:16] == b"BC\x02\x00", (p, h)
        total = int.from_bytes(h[16:18], "little") + 1
        blk = stream[p:p + total]
        assert len(blk) == total, (p, total, len(blk))
        out.append((blk[18:-8], int.from_bytes(blk[-4:], "little"), int.from_bytes(blk[-8:-4], "little")))
        p += total
    return out


class Unpack:
    """msx_unpack: the record walk on the device (inflated BAM bytes in, msx_batch view out)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.msx_unpack_create(ctx.h, C.byref(h)))
        self.h = h
        self._keep = None

    def seed(self, carry=b"", prev_name=None):
        buf = (C.c_uint8 * max(len(carry), 1)).from_buffer_copy(bytes(carry) or b"\0")
        self.ctx.check(self.ctx.lib.msx_unpack_seed(self.ctx.h, self.h, buf, len(carry),
                                                    prev_name.encode() if isinstance(prev_name, str) else prev_name))

    def prefetch(self, data):
        """send the bytes of the next enqueue ahead (msx_unpack_prefetch); pass the same `data` to enqueue"""
        data = bytes(data)
        self._pre = (data, (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0"))
        self.ctx.check(self.ctx.lib.msx_unpack_prefetch(self.ctx.h, self.h, self._pre[1], len(data)))

    def enqueue(self, data, pool_mode=0, want_aux=True, want_stats=True, n_targets=1 << 30, last=False, cut_mapped=False,
                unmapped_visible=False):
        data = bytes(data)
        pre = getattr(self, "_pre", None)
        if pre is not None and pre[0] == data:
            self._keep = pre[1]              # the buffer already on its way
        else:
            self._keep = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
        self._pre = None
        prm = L.UnpackParams(pool_mode, int(unmapped_visible), int(want_aux), int(want_stats), int(n_targets), int(last),
                             int(cut_mapped), 0)
        self.ctx.check(self.ctx.lib.msx_unpack_enqueue(self.ctx.h, self.h, self._keep, len(data), C.byref(prm)))

    def _prm(self, pool_mode, want_aux, want_stats, n_targets, last, cut_mapped, unmapped_visible):
        return L.UnpackParams(pool_mode, int(unmapped_visible), int(want_aux), int(want_stats), int(n_targets), int(last),
                              int(cut_mapped), 0)

    def prefetch_bgzf(self, comp, blocks, n_blocks):
        """msx_unpack_prefetch_bgzf; pass the object this returns to enqueue_bgzf"""
        buf = (C.c_uint8 * max(len(comp), 1)).from_buffer_copy(bytes(comp) or b"\0")
        self.ctx.check(self.ctx.lib.msx_unpack_prefetch_bgzf(self.ctx.h, self.h, buf, len(comp), blocks, n_blocks))
        return (buf, len(comp), blocks, n_blocks)

    def enqueue_bgzf(self, comp, blocks=None, n_blocks=None, pool_mode=0, want_aux=True, want_stats=True, n_targets=1 << 30,
                     last=False, cut_mapped=False, unmapped_visible=False):
        """the batch's new bytes as BGZF payloads (api.bgzf_blocks); `comp` may be what prefetch_bgzf returned"""
        if isinstance(comp, tuple):
            buf, n, blocks, n_blocks = comp
        else:
            buf, n = (C.c_uint8 * max(len(comp), 1)).from_buffer_copy(bytes(comp) or b"\0"), len(comp)
        self._keep = (buf, blocks)
        prm = self._prm(pool_mode, want_aux, want_stats, n_targets, last, cut_mapped, unmapped_visible)
        self.ctx.check(self.ctx.lib.msx_unpack_enqueue_bgzf(self.ctx.h, self.h, buf, n, blocks, n_blocks, C.byref(prm)))

    def finish(self):
        """(UnpackResult, DeviceBatch view -- owned by the unpacker: do not free)"""
        res = L.UnpackResult()
        b = L.Batch()
        self.ctx.check(self.ctx.lib.msx_unpack_finish(self.ctx.h, self.h, C.byref(res), C.byref(b)))
        view = DeviceBatch(self.ctx, b)
        view.free = lambda: None
        return res, view

    def offsets(self, n_records):
        out = np.zeros(n_records + 1, np.uint32)
        self.ctx.check(self.ctx.lib.msx_unpack_offsets(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p), n_records + 1))
        return out

    def emit(self, emit_ptr, n_emit, cap):
        out = np.zeros(max(cap, 1), np.uint8)
        nb = C.c_int64(0)
        self.ctx.check(self.ctx.lib.msx_unpack_emit(self.ctx.h, self.h, C.c_void_p(emit_ptr), int(n_emit),
                                                    out.ctypes.data_as(C.c_void_p), out.size, C.byref(nb)))
        return out[:nb.value].tobytes()

    def emit_bgzf(self, emit_ptr, n_emit, cap, level=0):
        """filter's output as finished BGZF blocks (msx_unpack_emit_gather_bgzf + fetch)"""
        nb, nblk = C.c_int64(0), C.c_int64(0)
        self.ctx.check(self.ctx.lib.msx_unpack_emit_gather_bgzf(self.ctx.h, self.h, C.c_void_p(emit_ptr), int(n_emit), level,
                                                               C.byref(nb), C.byref(nblk)))
        out = np.zeros(max(nb.value, 1), np.uint8)
        self.ctx.check(self.ctx.lib.msx_unpack_emit_fetch(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p), out.size, None))
        return out[:nb.value].tobytes(), nblk.value

    def emit_bgzf_enqueue(self, emit_ptr, n_emit, level=6):
        """hand the batch's output records to the encoder on its own stream (msx_unpack_emit_bgzf_enqueue)"""
        self.ctx.check(self.ctx.lib.msx_unpack_emit_bgzf_enqueue(self.ctx.h, self.h, C.c_void_p(emit_ptr), int(n_emit), level))

    def emit_bgzf_complete(self):
        """the oldest enqueued batch's blocks (msx_unpack_emit_bgzf_complete + fetch): (bytes, number of blocks)"""
        nb, nblk = C.c_int64(0), C.c_int64(0)
        self.ctx.check(self.ctx.lib.msx_unpack_emit_bgzf_complete(self.ctx.h, self.h, C.byref(nb), C.byref(nblk)))
        out = np.zeros(max(nb.value, 1), np.uint8)
        self.ctx.check(self.ctx.lib.msx_unpack_emit_fetch(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p), out.size, None))
        return out[:nb.value].tobytes(), nblk.value

    def close(self):
        if self.h:
            self.ctx.lib.msx_unpack_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class FilterResult:
    def __init__(self, keep, emit, as_out, n_emit):
        self.keep, self.emit, self.as_out, self.n_emit = keep, emit, as_out, n_emit


class FilterRun:
    """Device buffers + launch of one `filter` pass over a device batch."""

    def __init__(self, ctx, batch, want_emit=True, **opts):
        self.ctx, self.batch = ctx, batch
        self.fp = opts.pop("params", None) or filter_params(**opts)
        n = max(batch.n_records, 1)
        self.keep = ctx.alloc(n)
        self.emit = ctx.alloc(4 * n) if want_emit else None
        self.as_out = ctx.alloc(4 * n) if self.fp.rescore else None
        self.out = L.FilterOut(self.keep, self.emit, self.as_out)

    def enqueue(self):
        c = self.ctx
        c.check(c.lib.msx_filter_enqueue(c.h, C.byref(self.batch.b), C.byref(self.fp), C.byref(self.out)))

    def enqueue_with_profile(self, prof):
        """filter | profile in one call (msx_filter_profile_enqueue)."""
        c = self.ctx
        c.check(c.lib.msx_filter_profile_enqueue(c.h, C.byref(self.batch.b), C.byref(self.fp), C.byref(self.out),
                                                 prof.h))

    def finish(self):
        c = self.ctx
        st = L.FilterStatus()
        rc = c.lib.msx_filter_finish(c.h, C.byref(st))
        self.status = st
        c.check(rc)
        return st

    def result(self):
        c, n = self.ctx, self.batch.n_records
        ne = int(self.status.n_emit)
        keep = c.to_host(self.keep, n, np.uint8)
        emit = c.to_host(self.emit, ne, np.int32) if self.emit else None
        as_out = c.to_host(self.as_out, n, np.int32) if self.as_out else None
        return FilterResult(keep, emit, as_out, ne)

    def free(self):
        for p in (self.keep, self.emit, self.as_out):
            self.ctx.free(p)
        self.keep = self.emit = self.as_out = None


def run_filter(ctx, batch, **opts):
    """Enqueue + finish + fetch.  Raises MsxError with the reference's message on data errors."""
    run = FilterRun(ctx, batch, **opts)
    try:
        run.enqueue()
        run.finish()
        return run.result()
    finally:
        run.free()


def aln_stats(ctx, batch):
    n = max(batch.n_records, 1)
    ptrs = [ctx.alloc(4 * n) for _ in range(4)] + [ctx.alloc(n)]
    try:
        ctx.check(ctx.lib.msx_aln_stats(ctx.h, C.byref(batch.b), *[C.c_void_p(p) for p in ptrs]))
        names = ("length", "qlen", "qclip", "edit")
        out = {k: ctx.to_host(p, batch.n_records, np.int32) for k, p in zip(names, ptrs[:4])}
        out["status"] = ctx.to_host(ptrs[4], batch.n_records, np.uint8)
        return out
    finally:
        for p in ptrs:
            ctx.free(p)


class Profile:
    """msx_profile: insert counting + proportional sharing for one sample."""

    def __init__(self, ctx, n_features, multi="proportional", fmap=None):
        self.ctx = ctx
        self.n_features = int(n_features)
        self.multi = multi
        h = C.c_void_p()
        fm, nt = None, 0
        if fmap is not None:
            fmap = np.ascontiguousarray(fmap, dtype=np.int32)
            fm, nt = fmap.ctypes.data_as(C.c_void_p), int(fmap.size)
        ctx.check(ctx.lib.msx_profile_create(ctx.h, C.byref(h), self.n_features, L.MULTI[multi], fm, nt))
        self.h = h

    def reset(self):
        self.ctx.check(self.ctx.lib.msx_profile_reset(self.ctx.h, self.h))

    def accumulate(self, batch, keep_ptr=None):
        c = self.ctx
        c.check(c.lib.msx_profile_accumulate(c.h, self.h, C.byref(batch.b),
                                             C.c_void_p(keep_ptr) if keep_ptr else None))

    def merge(self, other):
        """Adds what `other` (a Profile of the same sample on any context of this process) has counted."""
        self.ctx.check(self.ctx.lib.msx_profile_merge(self.ctx.h, self.h, other.ctx.h, other.h))

    def accumulators(self):
        ui, d, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        c = self.ctx
        c.check(c.lib.msx_profile_accumulators(c.h, self.h, C.byref(ui), C.byref(d), C.byref(cnt)))
        return ui.value, d.value, cnt.value

    def prop_begin(self):
        self.ctx.check(self.ctx.lib.msx_profile_prop_begin(self.ctx.h, self.h))

    def prop_local(self):
        inc = C.c_void_p()
        self.ctx.check(self.ctx.lib.msx_profile_prop_local(self.ctx.h, self.h, C.byref(inc)))
        return inc.value

    def prop_local_slice(self, slice_, n_slices):
        """-> (device pointer of share[], first feature, count): the range this slice has completed"""
        inc, first, count = C.c_void_p(), C.c_int32(0), C.c_int32(0)
        self.ctx.check(self.ctx.lib.msx_profile_prop_local_slice(self.ctx.h, self.h, slice_, n_slices, C.byref(inc), C.byref(first), C.byref(count)))
        return inc.value, first.value, count.value

    def prop_apply(self):
        d = C.c_double(0)
        self.ctx.check(self.ctx.lib.msx_profile_prop_apply(self.ctx.h, self.h, C.byref(d)))
        return d.value

    def share_ptr(self):
        """Device pointer of the vector all-reduced per iteration (valid after create)."""
        inc = C.c_void_p()
        self.ctx.check(self.ctx.lib.msx_profile_share_dev(self.ctx.h, self.h, C.byref(inc)))
        return inc.value

    def prop_apply_enqueue(self):
        self.ctx.check(self.ctx.lib.msx_profile_prop_apply_enqueue(self.ctx.h, self.h))

    def prop_purged_enqueue(self):
        d = C.c_void_p()
        self.ctx.check(self.ctx.lib.msx_profile_prop_purged_enqueue(self.ctx.h, self.h, C.byref(d)))
        return d.value

    def prop_purged(self):
        v = C.c_uint32(0)
        self.ctx.check(self.ctx.lib.msx_profile_prop_purged(self.ctx.h, self.h, C.byref(v)))
        return v.value

    def abundance_ptr(self):
        a = C.c_void_p()
        self.ctx.check(self.ctx.lib.msx_profile_abundance_dev(self.ctx.h, self.h, C.byref(a)))
        return a.value

    def multi_size(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx.check(self.ctx.lib.msx_profile_multi_size(self.ctx.h, self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def shared_size(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.ctx.check(self.ctx.lib.msx_profile_shared_size(self.ctx.h, self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def finalize_enqueue(self):
        self.ctx.check(self.ctx.lib.msx_profile_finalize_enqueue(self.ctx.h, self.h))

    def finalize_dist_enqueue(self):
        """mInsertCountToAbundanceMatrix over all ranks' shards (RCCL inside the library)."""
        self.ctx.check(self.ctx.lib.msx_profile_finalize_dist_enqueue(self.ctx.h, self.h))

    def allreduce_counts(self):
        self.ctx.check(self.ctx.lib.msx_profile_allreduce_counts(self.ctx.h, self.h))

    def fetch(self):
        ab = np.zeros(max(self.n_features, 1), np.float64)
        st = L.ProfileStats()
        c = self.ctx
        c.check(c.lib.msx_profile_fetch(c.h, self.h, ab.ctypes.data_as(C.c_void_p), C.byref(st)))
        return ab[:self.n_features], st

    def finalize(self):
        self.finalize_enqueue()
        return self.fetch()

    def ui(self):
        ui, _, _ = self.accumulators()
        return self.ctx.to_host(ui, self.n_features, np.uint32)

    def close(self):
        if self.h:
            self.ctx.lib.msx_profile_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class RecordSlice:
    """Records [lo, hi) of a host SoA (tests/samio.Records, HostSynth) as a batch of their own: the coverage fields."""

    def __init__(self, rec, lo, hi):
        self.flag = rec.flag[lo:hi]
        self.tid = rec.tid[lo:hi]
        self.pos = rec.pos[lo:hi]
        co = np.asarray(rec.cigar_off[lo:hi + 1], dtype=np.int64)
        self.cigar_off = (co - co[0]).astype(np.uint32)
        self.cigar = rec.cigar[int(co[0]):int(co[-1])]
        self.n_records = hi - lo


def coverage_collected(ctx, batches, target_len, covered=False):
    """A sample that arrives batch after batch (the command line's loop): msx_coverage_collect per device batch, then
    msx_coverage_collect_finish.  Returns (depths per target, batches that took the streamed way[, covered flags])."""
    off = np.zeros(len(target_len) + 1, np.int64)
    off[1:] = np.cumsum(np.asarray(target_len, dtype=np.int64))
    total = int(off[-1])
    d_off = ctx.alloc(off.nbytes)
    d_cov = ctx.alloc(4 * max(total, 1) + 8)
    d_flag = ctx.alloc(len(target_len) + 8) if covered else None
    try:
        ctx.to_dev(d_off, off)
        ctx.to_dev(d_cov, np.full(max(total, 1) + 2, 0x5a5a5a5a, np.uint32))      # (garbage: the calls must not rely on zeros)
        if covered:
            ctx.zero(d_flag, len(target_len) + 8)
        for b in batches:
            ctx.check(ctx.lib.msx_coverage_collect(ctx.h, C.byref(b.b), C.c_void_p(d_off), len(target_len), total, C.c_void_p(d_cov),
                                                   C.c_void_p(d_flag) if covered else None))
        n_streamed = C.c_int64(-1)
        ctx.check(ctx.lib.msx_coverage_collect_finish(ctx.h, C.c_void_p(d_cov), total, C.byref(n_streamed)))
        ctx.sync()
        cov = ctx.to_host(d_cov, total, np.int32)
        flags = ctx.to_host(d_flag, len(target_len), np.uint8) if covered else None
    finally:
        ctx.free(d_off)
        ctx.free(d_cov)
        if covered:
            ctx.free(d_flag)
    per = [cov[off[i]:off[i + 1]] for i in range(len(target_len))]
    return (per, int(n_streamed.value), flags) if covered else (per, int(n_streamed.value))


def coverage(ctx, batch, target_len, summary=False, whole_sample=False):
    """Per-base depth per target (msam_coverage.c:33-87) for one device batch; summary=True: also (touched positions,
    depth sum) per target as the device takes them (msx_coverage_summary, msam_coverage.c:188-219);
    whole_sample=True: msx_coverage_depths (the batch is the sample: depths written once, no zeroing, no finish)."""
    off = np.zeros(len(target_len) + 1, np.int64)
    off[1:] = np.cumsum(np.asarray(target_len, dtype=np.int64))
    total = int(off[-1])
    d_off = ctx.alloc(off.nbytes)
    d_cov = ctx.alloc(4 * max(total, 1) + 8)
    try:
        ctx.to_dev(d_off, off)
        if whole_sample:
            ctx.to_dev(d_cov, np.full(max(total, 1) + 2, 0x5a5a5a5a, np.uint32))      # (garbage: the call must not rely on zeros)
            ctx.check(ctx.lib.msx_coverage_depths(ctx.h, C.byref(batch.b), C.c_void_p(d_off), len(target_len), total,
                                                  C.c_void_p(d_cov), None))
        else:
            ctx.zero(d_cov, 4 * max(total, 1) + 8)
            ctx.check(ctx.lib.msx_coverage_accumulate(ctx.h, C.byref(batch.b), C.c_void_p(d_off),
                                                      len(target_len), total, C.c_void_p(d_cov), None))
            ctx.check(ctx.lib.msx_coverage_finish(ctx.h, C.c_void_p(d_cov), total))
        cov = ctx.to_host(d_cov, total, np.int32)
        if summary:
            touched = np.zeros(max(len(target_len), 1), np.int64)
            dsum = np.zeros(max(len(target_len), 1), np.int64)
            ctx.check(ctx.lib.msx_coverage_summary(ctx.h, C.c_void_p(d_cov), C.c_void_p(d_off), len(target_len),
                                                   touched.ctypes.data_as(C.c_void_p), dsum.ctypes.data_as(C.c_void_p)))
    finally:
        ctx.free(d_off)
        ctx.free(d_cov)
    per = [cov[off[i]:off[i + 1]] for i in range(len(target_len))]
    return (per, touched[:len(target_len)], dsum[:len(target_len)]) if summary else per
