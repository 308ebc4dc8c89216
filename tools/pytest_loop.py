#!/usr/bin/env python3
"""The whole `-m gpu` suite N times over in ONE process (round 4's device fault appeared only in long-lived
processes -- heap churn of hundreds of megabytes of frame arrays beside the transfers -- and tests/conftest.py no longer fixes glibc's
mmap threshold).  pytest_loop.py [N]; exit status 0 iff every session passed.  Prints one line per session."""
import os
import sys
import time

import pytest

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
for k in range(n):
    t0 = time.time()
    rc = pytest.main([os.path.join(root, "tests"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"])
    print(f"[pytest_loop] session {k + 1} of {n}: exit {int(rc)} after {time.time() - t0:.0f} s (pid {os.getpid()})", flush=True)
    bad += int(rc) != 0
sys.exit(1 if bad else 0)
