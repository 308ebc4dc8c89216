#!/usr/bin/env python3
"""Aggregate resident throughput with N contexts in flight, each encoding its own resident copy of the batch (what a streaming
host does with several workers): two_in_flight.py [period qp nframes name nctx p_groups]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
period = int(sys.argv[1]) if len(sys.argv) > 1 else 0
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
name = sys.argv[4] if len(sys.argv) > 4 else "foremanlike"
nctx = int(sys.argv[5]) if len(sys.argv) > 5 else 2
if len(sys.argv) > 6:
    os.environ["ICSP_P_GROUPS"] = sys.argv[6]
clip = clipgen.synth_clip(name, n)
encs = [capi.Encoder(352, 288, qp, qp, period, max_frames=n) for _ in range(nctx)]
for e in encs:
    e.upload(clip)
for _ in range(50):
    for e in encs:
        e.encode_resident(0, n)
for e in encs:
    e.sync()
best = 0
passes = 200
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(passes):
        for e in encs:
            e.encode_resident(0, n)
    for e in encs:
        e.sync()
    dt = time.perf_counter() - t0
    best = max(best, passes * nctx * n / dt)
print(f"period={period} qp={qp} n={n} {name} contexts={nctx} p_groups={os.environ.get('ICSP_P_GROUPS', 'default')}: {best:10.0f} fps aggregate")
for e in encs:
    e.close()
