"""msamtools_amd -- MI355X-native msamtools filter -> profile hot path.

The compute lives in libmsamtools_amd.so (hand-written HIP for gfx950 behind
the C ABI of include/msamtools_amd.h).  Importing the package binds the
library and fails loudly if it has not been built; creating a Context fails
loudly without a gfx950 GPU.  There is no CPU fallback.
"""
from . import _lib
from ._lib import MsxError

_lib.load()

from .api import (Context, DeviceBatch, FilterRun, HostSynth, Profile, RecordSlice, Unpack, aln_stats,  # noqa: E402
                  bgzf_blocks, bgzf_deflate, bgzf_inflate, bgzf_split, coverage, coverage_collected, dist_unique_id, filter_params,
                  run_filter)
from .grouping import filter_pools, profile_pools  # noqa: E402

__all__ = ["Context", "DeviceBatch", "FilterRun", "HostSynth", "Profile", "Unpack", "MsxError", "aln_stats",
           "bgzf_blocks", "bgzf_deflate", "bgzf_inflate", "bgzf_split", "coverage", "coverage_collected", "RecordSlice", "dist_unique_id", "filter_params", "run_filter", "filter_pools", "profile_pools"]
