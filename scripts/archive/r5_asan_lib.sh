#!/bin/bash
# the library's host code under AddressSanitizer on the GPU box (scripts/archive/build_asan_lib.sh), through the python tests and the
# command line; gcc's libasan is preloaded for python; the command line is msamtools-asan (make asan), which links it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5asanlib; rm -rf $OUT; mkdir -p $OUT
RT=$(readlink -f $(gcc -print-file-name=libasan.so))
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0:abort_on_error=1:alloc_dealloc_mismatch=0
export MSX_LIB_PATH=$PWD/build/asanlib/libmsamtools_amd.so LD_LIBRARY_PATH=$PWD/build/asanlib:$LD_LIBRARY_PATH
timeout 120 msamtools_amd/bin/msamtools-asan filter -l 80 -p 95 -z 80 --besthit -S tests/golden/fixtures/besthit.sam > $OUT/tiny.out 2> $OUT/tiny.err; echo "tiny rc=$? lines=$(wc -l < $OUT/tiny.out)"; tail -5 $OUT/tiny.err
ldd msamtools_amd/bin/msamtools-asan | grep msamtools_amd
# (torch's amdsmi start-up and RCCL's ncclGetUniqueId abort in a python with a preloaded ASan runtime -- without a report, not in
#  this library: the tests that import torch or make a communicator from python stay out; the command line's one-rank
#  communicator test runs, msamtools-asan links the runtime instead of preloading it)
LD_PRELOAD=$RT timeout 2000 python -m pytest --timeout=600 -q -m gpu ${TESTS:-tests} --ignore=tests/test_gpu_two_ranks.py \
  --deselect tests/test_gpu_scale.py::test_bench_multi_gpu_step_on_one_rank -k 'not test_one_rank_distributed_finalize_equals_plain_and_oracle' > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
grep -c "AddressSanitizer" $OUT/pytest.log
