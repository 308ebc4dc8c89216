#!/usr/bin/env python3
"""Resident throughput with the caller rotating over R disjoint resident ranges of n frames each (what a streaming host issues;
bench.py's regime is R = 2): alt_ranges.py [period qp n nranges passes [width height]].  Environment (ICSP_WHOLE, ICSP_P_GROUPS, ...) is echoed.
ICSP_ALT_MANY=k: the R ranges form R / k lists of k ranges each; a call hands one LIST to the library (icsp_encode_resident_many), the
calls rotate over the lists (k = R: one list again and again, every pass following the one before; k = R / 2: two lists alternating)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
period = int(sys.argv[1]) if len(sys.argv) > 1 else 0
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 200
W = int(sys.argv[6]) if len(sys.argv) > 6 else 352
H = int(sys.argv[7]) if len(sys.argv) > 7 else 288
name = "stefanlike" if period else "foremanlike"
enc = capi.Encoder(W, H, qp, qp, period, max_frames=n * R)
for r in range(R):
    if (W, H) == (352, 288):
        enc.upload(clipgen.synth_clip(name, n, first_frame=(r * n) % 600), first=r * n)
    else:                                   # (large frames: a short clip repeated, the generator is slow)
        base = clipgen.synth_clip(name, min(n, 10), width=W, height=H, first_frame=r)
        for f in range(0, n, len(base)):
            enc.upload(base[:min(len(base), n - f)], first=r * n + f)
MANY = int(os.environ.get("ICSP_ALT_MANY", "0"))
LISTS = [[(r * n, n) for r in range(j, R, max(1, R // MANY))] for j in range(max(1, R // MANY))] if MANY else []


def one(k):
    if MANY:
        enc.encode_resident_many(LISTS[k % len(LISTS)])
    else:
        enc.encode_resident((k % R) * n, n)


for k in range(100 // (MANY if MANY else 1) + 2):
    one(k)
enc.sync()
best = 0
host = 1.0
for rep in range(3):
    t0 = time.perf_counter()
    calls = max(1, passes // MANY) if MANY else passes
    for k in range(calls):
        one(k)
    t1 = time.perf_counter()
    enc.sync()
    dt = (time.perf_counter() - t0) / calls
    if (n * MANY if MANY else n) / dt > best:
        best = (n * MANY if MANY else n) / dt
        host = (t1 - t0) / (dt * calls)            # share of the region the host spent issuing the calls (near 1: the host is what paces it)
print(f"host {host:.2f} ", end="")
print(f"{W}x{H} " * ((W, H) != (352, 288)) + f"period={period} qp={qp} n={n} ranges={R}: {best:10.0f} fps  ({n / best * 1e3:.4f} ms/step) env={ {k: v for k, v in os.environ.items() if k.startswith(('HIP_', 'ICSP_', 'GPU_'))} }")
enc.close()
