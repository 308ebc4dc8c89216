#!/usr/bin/env python3
"""Does an idle context's streams cost a busy context its hardware queues?  exp_two_ctx.py none|alive|closed
An all-intra context is created and used first (three streams), then the IPPP alternating regime is measured on a second one
with the first left alone / still alive / closed before."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, clipgen
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
n = 300
enc0 = None
if mode != "none":
    enc0 = capi.Encoder(352, 288, 16, 16, 0, max_frames=2 * n)
    c = clipgen.synth_clip("foremanlike", n)
    enc0.upload(c, first=0); enc0.upload(c, first=n)
    for k in range(20):
        enc0.encode_resident((k & 1) * n, n)
    enc0.sync()
    if mode == "closed":
        enc0.close(); enc0 = None
enc = capi.Encoder(352, 288, 8, 8, 10, max_frames=2 * n)
for r in range(2):
    enc.upload(clipgen.synth_clip("stefanlike", n, first_frame=r * n), first=r * n)
for k in range(60):
    enc.encode_resident((k & 1) * n, n)
enc.sync()
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(200):
        enc.encode_resident((k & 1) * n, n)
    enc.sync()
    best = max(best, 200 * n / (time.perf_counter() - t0))
print(f"first context {mode:7s} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')}: IPPP two ranges alternating {best:10.0f} fps")
enc.close()
if enc0: enc0.close()
