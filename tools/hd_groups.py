#!/usr/bin/env python3
"""1920x1088, 3000 frames = 100 closed GOPs of 30, QP 16 (BASELINE configs[4]) on one GPU: frames/s of a resident pass.
Environment (ICSP_P_GROUPS ...) is echoed: hd_groups.py [frames] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, clipgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h = 1920, 1088
base = clipgen.synth_clip("tablelike", 12, width=w, height=h)
enc = capi.Encoder(w, h, 16, 16, 30, max_frames=n)
for f in range(0, n, 12):
    enc.upload(base[:min(12, n - f)], first=f)
enc.encode_resident(0, n); enc.sync()
best = 0
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(passes): enc.encode_resident(0, n)
    enc.sync()
    best = max(best, n * passes / (time.perf_counter() - t0))
print(f"1088p n={n}: {best:9.1f} fps  choice={enc.last_choice()} env={ {k: v for k, v in os.environ.items() if k.startswith('ICSP_')} }")
enc.close()
