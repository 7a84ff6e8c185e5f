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


@pytest.fixture(scope="session")
def expectations():
    import json
    with open(os.path.join(GOLDEN, "reference_expectations.json")) as fh:
        return json.load(fh)


def fixture_path(name):
    return os.path.join(FIXTURES, name)
