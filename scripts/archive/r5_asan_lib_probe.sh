#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0:alloc_dealloc_mismatch=0
export MSX_LIB_PATH=$PWD/build/asanlib/libmsamtools_amd.so LD_LIBRARY_PATH=$PWD/build/asanlib:$LD_LIBRARY_PATH
LD_PRELOAD=$RT timeout 120 python3 -X faulthandler -c "
import msamtools_amd as m
print('imported', flush=True)
c = m.Context(0)
print('ok', flush=True)
" 2>&1 | tail -40
