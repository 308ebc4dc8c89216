#!/usr/bin/env python3
"""How long does the HOST take to enqueue one resident encode pass (no sync inside), against the pass's GPU time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, clipgen
for name, q, period in (("foremanlike", 16, 0), ("stefanlike", 8, 10)):
    clip = clipgen.synth_clip(name, 300)
    enc = capi.Encoder(352, 288, q, q, period, max_frames=300)
    enc.upload(clip)
    for _ in range(50):
        enc.encode_resident(0, 300)
    enc.sync()
    host = []
    for _ in range(20):
        enc.sync()
        t0 = time.perf_counter(); enc.encode_resident(0, 300); host.append(time.perf_counter() - t0)
        enc.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        enc.encode_resident(0, 300)
    enc.sync()
    dt = (time.perf_counter() - t0) / 100
    host.sort()
    print(f"period {period}: host enqueue of one pass {host[len(host)//2]*1e6:.0f} us (min {host[0]*1e6:.0f}); pass back-to-back {dt*1e6:.0f} us")
    enc.close()
