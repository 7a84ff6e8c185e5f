"""ctypes binding of libmsamtools_amd.so (the C ABI in include/msamtools_amd.h).

The library is built in-tree (msamtools_amd/libmsamtools_amd.so) by
`make -C msamtools_amd/csrc` / __graft_entry__.build().  There is no Python or
CPU fallback: if the shared object is missing, or no gfx950 device is present
when a context is created, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MSX_LIB_PATH: another build of the same library -- scripts/archive/build_asan_lib.sh (a one-off of round 5, not part of the routine gate) puts one with its host code under
#  AddressSanitizer into build/asanlib; there is no other implementation to point it at)
LIB_PATH = os.environ.get("MSX_LIB_PATH") or os.path.join(_HERE, "libmsamtools_amd.so")

MSX_OK = 0
ERR_NO_MD_NM, ERR_NO_AS, ERR_NO_FILTER, ERR_SHARE_TYPE = 1, 2, 3, 4
ERR_HIP, ERR_ARG, ERR_NOMEM, ERR_NO_DEVICE = -10, -11, -12, -13
HAS_MD, HAS_NM, HAS_AS = 1, 2, 4
MULTI = {"all": 1, "equal": 2, "proportional": 3, "ignore": 4}


class MsxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"[msx {code}] {text}")
        self.code = code
        self.text = text


class Batch(C.Structure):
    _fields_ = [
        ("n_records", C.c_int64), ("n_groups", C.c_int64),
        ("flag", C.c_void_p), ("rflags", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p),
        ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("md_off", C.c_void_p), ("md", C.c_void_p),
        ("nm", C.c_void_p), ("as_", C.c_void_p), ("group_off", C.c_void_p), ("qname_hash", C.c_void_p),
        ("pool_rule", C.c_int32), ("reserved_", C.c_int32),
    ]


POOLS_PROFILE, POOLS_FILTER = 0, 1


class FilterParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("min_length", "ppt", "max_clip", "rescore", "invert", "keep_unmapped", "besthit", "uniqhit",
                 "fatal_pool_partial")]


class FilterOut(C.Structure):
    _fields_ = [("keep", C.c_void_p), ("emit_idx", C.c_void_p), ("as_out", C.c_void_p)]


class FilterStatus(C.Structure):
    _fields_ = [("n_emit", C.c_int64), ("err_record", C.c_int64)]


class ProfileStats(C.Structure):
    _fields_ = [
        ("insert_count", C.c_uint32), ("uniq_mapper_count", C.c_uint32),
        ("multi_mapper_count", C.c_uint32), ("purged_insert_count", C.c_uint32),
        ("iterations", C.c_int32), ("converged", C.c_int32), ("delta", C.c_double * 20),
    ]


class UnpackParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("pool_mode", "unmapped_visible", "want_aux", "want_stats", "n_targets", "last",
                                         "cut_mapped", "reserved_")]


class BgzfBlock(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32),
                ("crc32", C.c_uint32), ("reserved_", C.c_uint32)]


class UnpackResult(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_records", "n_groups", "bytes_consumed", "carry_bytes", "bad_guesses")]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_groups", C.c_int64), ("n_refs", C.c_int32),
                ("mean_extra_hits", C.c_int32), ("first_group", C.c_int64)]


class SynthSizes(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("n_cigar", C.c_int64), ("n_md", C.c_int64)]


DIST_ID_BYTES = 128

# every symbol include/msamtools_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "msx_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "msx_ctx_destroy": (None, [_P]),
    "msx_last_error": (C.c_char_p, [_P]),
    "msx_abi_version": (C.c_int, []),
    "msx_debug_guard_check": (C.c_int64, []),
    "msx_debug_guard_selftest": (C.c_int64, [C.c_int]),
    "msx_runtime_warmup": (C.c_int, [C.c_int]),
    "msx_ctx_stream": (_P, [_P]),
    "msx_ctx_sync": (C.c_int, [_P]),
    "msx_batch_upload": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(Batch)]),
    "msx_batch_free": (None, [_P, C.POINTER(Batch)]),
    "msx_stage_create": (C.c_int, [_P, C.POINTER(_P)]),
    "msx_stage_destroy": (None, [_P, _P]),
    "msx_stage_upload": (C.c_int, [_P, _P, C.POINTER(Batch), C.POINTER(Batch)]),
    "msx_stage_outputs": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.POINTER(FilterOut)]),
    "msx_host_alloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "msx_host_free": (None, [_P, _P]),
    "msx_host_register": (C.c_int, [_P, _P, C.c_size_t]),
    "msx_host_unregister": (C.c_int, [_P, _P]),
    "msx_dev_to_host_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msx_event_create": (C.c_int, [_P, C.POINTER(_P)]),
    "msx_event_record": (C.c_int, [_P, _P]),
    "msx_event_wait": (C.c_int, [_P, _P]),
    "msx_event_destroy": (None, [_P, _P]),
    "msx_unpack_create": (C.c_int, [_P, C.POINTER(_P)]),
    "msx_unpack_destroy": (None, [_P, _P]),
    "msx_unpack_seed": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_char_p]),
    "msx_unpack_carry": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.POINTER(C.c_int)]),
    "msx_unpack_enqueue": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(UnpackParams)]),
    "msx_unpack_finish": (C.c_int, [_P, _P, C.POINTER(UnpackResult), C.POINTER(Batch)]),
    "msx_unpack_prefetch": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msx_unpack_emit": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_int64)]),
    "msx_unpack_offsets": (C.c_int, [_P, _P, _P, C.c_int64]),
    "msx_bgzf_inflate": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int64, _P, _P, _P]),
    "msx_unpack_enqueue_bgzf": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_int64, _P]),
    "msx_unpack_emit_gather": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "msx_unpack_emit_fetch": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "msx_unpack_emit_gather_bgzf": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P]),
    "msx_unpack_emit_bgzf_enqueue": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int]),
    "msx_unpack_emit_bgzf_complete": (C.c_int, [_P, _P, _P, _P]),
    "msx_bgzf_bound": (C.c_int64, [C.c_int64, C.c_int]),
    "msx_bgzf_deflate": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P, C.c_size_t, _P, _P]),
    "msx_unpack_prefetch_bgzf": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.c_int64]),
    "msx_filter_enqueue": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(FilterParams), C.POINTER(FilterOut)]),
    "msx_filter_finish": (C.c_int, [_P, C.POINTER(FilterStatus)]),
    "msx_aln_stats": (C.c_int, [_P, C.POINTER(Batch), _P, _P, _P, _P, _P]),
    "msx_profile_create": (C.c_int, [_P, C.POINTER(_P), C.c_int32, C.c_int32, _P, C.c_int32]),
    "msx_profile_destroy": (None, [_P, _P]),
    "msx_profile_reset": (C.c_int, [_P, _P]),
    "msx_profile_accumulate": (C.c_int, [_P, _P, C.POINTER(Batch), _P]),
    "msx_filter_profile_enqueue": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(FilterParams), C.POINTER(FilterOut), _P]),
    "msx_profile_merge": (C.c_int, [_P, _P, _P, _P]),
    "msx_profile_accumulators": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "msx_profile_prop_begin": (C.c_int, [_P, _P]),
    "msx_profile_prop_local": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "msx_ctx_set_lanes": (C.c_int, [_P, C.c_int]),
    "msx_profile_prop_local_slice": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "msx_profile_prop_apply": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "msx_profile_prop_purged": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "msx_profile_prop_apply_enqueue": (C.c_int, [_P, _P]),
    "msx_profile_share_dev": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "msx_profile_prop_purged_enqueue": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "msx_profile_finalize": (C.c_int, [_P, _P, _P, C.POINTER(ProfileStats)]),
    "msx_profile_finalize_enqueue": (C.c_int, [_P, _P]),
    "msx_profile_fetch": (C.c_int, [_P, _P, _P, C.POINTER(ProfileStats)]),
    "msx_profile_abundance_dev": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "msx_profile_multi_size": (C.c_int, [_P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "msx_profile_shared_size": (C.c_int, [_P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "msx_coverage_accumulate": (C.c_int, [_P, C.POINTER(Batch), _P, C.c_int32, C.c_int64, _P, _P]),
    "msx_coverage_finish": (C.c_int, [_P, _P, C.c_int64]),
    "msx_coverage_depths": (C.c_int, [_P, C.POINTER(Batch), _P, C.c_int32, C.c_int64, _P, _P]),
    "msx_coverage_collect": (C.c_int, [_P, C.POINTER(Batch), _P, C.c_int32, C.c_int64, _P, _P]),
    "msx_coverage_collect_finish": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "msx_coverage_summary": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "msx_synth_device": (C.c_int, [_P, C.POINTER(SynthParams), C.POINTER(Batch), C.POINTER(SynthSizes)]),
    "msx_synth_host": (C.c_int, [C.POINTER(SynthParams), C.POINTER(Batch), C.POINTER(SynthSizes)]),
    "msx_synth_host_free": (None, [C.POINTER(Batch)]),
    "msx_dev_alloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "msx_dev_free": (None, [_P, _P]),
    "msx_dev_zero": (C.c_int, [_P, _P, C.c_size_t]),
    "msx_dev_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msx_host_to_dev": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msx_dist_unique_id": (C.c_int, [_P]),
    "msx_dist_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "msx_dist_init_env": (C.c_int, [_P]),
    "msx_dist_finalize": (None, [_P]),
    "msx_dist_rank": (C.c_int, [_P]),
    "msx_dist_world": (C.c_int, [_P]),
    "msx_dist_rendezvous": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, C.c_int]),
    "msx_dist_barrier": (C.c_int, [_P]),
    "msx_dist_max_f64": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "msx_dist_sum_i64": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "msx_profile_allreduce_counts": (C.c_int, [_P, _P]),
    "msx_profile_finalize_dist_enqueue": (C.c_int, [_P, _P]),
    "msx_timing_enable": (C.c_int, [_P, C.c_int]),
    "msx_timing_reset": (C.c_int, [_P]),
    "msx_timing_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "msx_timing_get_bytes": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """Load the shared object and bind every declared symbol (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C msamtools_amd/csrc` "
            "(hipcc --offload-arch=gfx950). msamtools_amd has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc):
    if rc != MSX_OK:
        text = load().msx_last_error(ctx_handle)
        raise MsxError(rc, text.decode() if text else "")
    return rc
