#!/usr/bin/env python3
"""Quick resident-throughput check (not the bench contract): quick.py [period qp nframes name passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
period = int(sys.argv[1]) if len(sys.argv) > 1 else 10
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
name = sys.argv[4] if len(sys.argv) > 4 else "stefanlike"
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 200
base = clipgen.synth_clip(name, min(n, 300))
clip = np.concatenate([base] * ((n + len(base) - 1) // len(base)))[:n]
enc = capi.Encoder(352, 288, qp, qp, period, max_frames=n)
enc.upload(clip)
for _ in range(100):
    enc.encode_resident(0, n)
enc.sync()
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(passes):
        enc.encode_resident(0, n)
    enc.sync()
    dt = (time.perf_counter() - t0) / passes
    best = max(best, n / dt)
print(f"period={period} qp={qp} n={n} {name}: {best:10.0f} fps  ({n / best * 1e3:.4f} ms/pass) env={ {k: v for k, v in os.environ.items() if k.startswith(('HIP_', 'ICSP_', 'GPU_', 'ROC_', 'AMD_'))} }")
enc.close()
