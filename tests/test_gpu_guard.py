"""MSX_GUARD=1 (msx_guard.hip): the guard bytes around the library's device allocations see a kernel's stray write, on either
side, and stay intact through a filter | profile call and a coverage call (-m gpu; in a child process: the switch is read once)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import msamtools_amd as m
c = m.Context(0)
assert c.lib.msx_debug_guard_check() == 0
assert c.lib.msx_debug_guard_selftest(0) == 1 and c.lib.msx_debug_guard_selftest(1) == 1
db = m.DeviceBatch.synth(c, 7, 30000, 500, 4)
prof = m.Profile(c, 500, "proportional")
run = m.FilterRun(c, db, l=80, p=95, z=80, besthit=True)
run.enqueue_with_profile(prof)
run.finish()
ab, st = prof.finalize()
assert st.insert_count > 0
cov = m.coverage(c, db, [5000] * 500, whole_sample=True)
assert sum(int(x.sum()) for x in cov) > 0
run.free(); prof.close(); db.free()
assert c.lib.msx_debug_guard_check() == 0
c.close()
print("guard ok")
"""


def test_guard_sees_stray_writes_and_a_clean_run_leaves_it_alone():
    env = dict(os.environ, MSX_GUARD="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"guard ok" in r.stdout, r.stderr.decode()[-1500:]
    assert r.stderr.count(b"msx_debug_guard_selftest: expected") == 2


def test_guard_is_off_by_default():
    env = {k: v for k, v in os.environ.items() if k != "MSX_GUARD"}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", "import msamtools_amd as m; c = m.Context(0); print(c.lib.msx_debug_guard_check(), c.lib.msx_debug_guard_selftest(0))"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == [b"-1", b"-1"], r.stderr.decode()[-500:]
