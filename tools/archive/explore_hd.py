#!/usr/bin/env python3
"""Ad-hoc: 1920x1088 throughput (BASELINE configs[4] geometry), per-kernel times.  usage: explore_hd.py [nframes] [period]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
period = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w, h = 1920, 1088
base = clipgen.synth_clip("tablelike", min(n, 12), width=w, height=h)
clip = np.concatenate([base] * ((n + len(base) - 1) // len(base)))[:n]
enc = capi.Encoder(w, h, 16, 16, period, max_frames=n)
enc.upload(clip)
enc.encode_resident(0, n); enc.sync()
enc.profile(True)
t0 = time.perf_counter()
for _ in range(3):
    enc.encode_resident(0, n)
enc.sync()
dt = (time.perf_counter() - t0) / 3
prof = {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in enc.profile_get().items() if v[1]}
enc.profile(False)
t0 = time.perf_counter(); enc.decode_resident(0, n); enc.sync(); ddt = time.perf_counter() - t0
enc.close()
mbps = n * w * h * 1.5 / dt / 1e9
print(f"1088p n={n} period={period}: {n/dt:9.1f} fps  {dt*1e3:8.2f} ms  input {mbps:.2f} GB/s  decode {n/ddt:9.1f} fps  {prof}")
