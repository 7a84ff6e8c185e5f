#!/bin/bash
# MSX_GUARD=1: 512 guard bytes around every device allocation of the library, checked at free, after every GPU test and at
# the command line's exit.  First that the guard sees an overrun at all (msx_debug_guard_selftest: a kernel's one byte behind / in front of an allocation), then
# the GPU suite under it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5guard; rm -rf $OUT; mkdir -p $OUT
export MSX_GUARD=1
timeout 120 python3 - > $OUT/selftest.log 2>&1 <<'PY'
import msamtools_amd as m
c = m.Context(0)
print("clean:", c.lib.msx_debug_guard_check())
print("a kernel's byte behind the allocation:", c.lib.msx_debug_guard_selftest(0), "; in front of it:", c.lib.msx_debug_guard_selftest(1))
PY
cat $OUT/selftest.log
timeout 2400 python -m pytest --timeout=600 -q -m gpu tests ${PYTEST_EXTRA} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
grep -c "MSX_GUARD" $OUT/pytest.log
