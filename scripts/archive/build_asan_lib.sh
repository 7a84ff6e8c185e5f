#!/bin/bash
# libmsamtools_amd.so with its HOST code under AddressSanitizer (device code untouched: -fno-gpu-sanitize; device-side ASan is
# not available on this pool) -> build/asanlib/libmsamtools_amd.so.  Used through MSX_LIB_PATH (python) / LD_LIBRARY_PATH (the
# command line) with gcc's libasan preloaded (python) or linked (msamtools-asan): scripts/archive/r5_asan_lib.sh.
set -e
cd "$(dirname "$0")/../../msamtools_amd/csrc"
OUT=../../build/asanlib
mkdir -p $OUT
FLAGS="-O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
for f in msx_*.hip; do
  ( hipcc $FLAGS -c $f -o $OUT/${f%.hip}.o ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
# linked WITHOUT a sanitizer runtime: the __asan_* symbols resolve against the runtime the process brings (gcc's libasan --
# preloaded for python, linked into msamtools-asan; ROCm's clang runtime intercepts HSA allocations for device-side ASan
# and fails on this pool: "out of memory: allocator is trying to allocate 0x400000 bytes" in hsa_amd_memory_pool_allocate)
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsamtools_amd.so $OUT/msx_*.o
ls -la $OUT/libmsamtools_amd.so
