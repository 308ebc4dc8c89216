import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (skipped elsewhere)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """Large numpy arrays as mappings of their own, always.  glibc raises its mmap threshold (up to 32 MB) after the first large
    free, so a long test session ends up allocating and trimming hundreds of megabytes of frame arrays on the brk heap; the HIP
    runtime pins host buffers of large plain-memory transfers on the fly, and a later transfer near a trimmed range faulted on the
    device ("Memory access fault by GPU ... Reason: Unknown", about one session in two in round 4, only after ~100 tests in one
    process and only with registered heap arrays in play; never in a fresh process).  A fixed threshold keeps every array of 128 KB
    or more in its own mapping, which is unmapped -- and its pins dropped -- when it is freed."""
    import ctypes
    try:
        ctypes.CDLL("libc.so.6").mallopt(-3, 128 * 1024)       # M_MMAP_THRESHOLD: setting it switches the dynamic adjustment off
    except Exception:
        pass
