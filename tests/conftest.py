import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")
FIXTURES = os.path.join(GOLDEN, "fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """MSX_TEST_ORDER=reverse | <seed>: the collected tests in reverse, or shuffled by the seed -- the contexts and their
    workspace are shared by the tests of a file, so an order nobody wrote down finds what depends on it (round 5: a path
    that read what an earlier, larger batch had left in a buffer)."""
    order = os.environ.get("MSX_TEST_ORDER")
    if not order:
        return
    if order == "reverse":
        items.reverse()
    else:
        import random
        random.Random(int(order)).shuffle(items)


def pytest_collection_finish(session):
    """Test files must not load the library while they are being collected: tests/test_gpu_two_ranks.py asks torch for
    the device count during collection, and a process that loads /opt/rocm's runtime first (through the library) and
    torch's bundled copy second ends up with two HSA runtimes -- RCCL then fails to initialise (seen as
    "pfn_hsa_system_get_info failed" / "no ROCm-capable device is detected" in every in-process RCCL test)."""
    if "msamtools_amd" in sys.modules:
        raise pytest.UsageError("a test module imported msamtools_amd during collection (import it inside tests and fixtures)")


@pytest.fixture(scope="session")
def expectations():
    import json
    with open(os.path.join(GOLDEN, "reference_expectations.json")) as fh:
        return json.load(fh)


def fixture_path(name):
    return os.path.join(FIXTURES, name)


@pytest.fixture(autouse=True)
def _device_guard_bytes(request):
    """MSX_GUARD=1: after every GPU test the guard bytes around the library's live device allocations must be intact
    (include/msamtools_amd.h: msx_debug_guard_check; freed allocations were checked when they were freed)."""
    yield
    if os.environ.get("MSX_GUARD") and request.node.get_closest_marker("gpu"):
        import msamtools_amd._lib as L
        bad = L.load().msx_debug_guard_check()
        assert bad <= 0, f"{bad} guard byte(s) around device allocations overwritten (stderr has the allocations)"
