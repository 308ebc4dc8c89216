import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: a wall-clock bound inside the GPU suite (retried once; -m gpu selects it, -m 'gpu and not perf' skips it)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (skipped elsewhere)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
