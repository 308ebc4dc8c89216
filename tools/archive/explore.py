#!/usr/bin/env python3
"""Ad-hoc throughput exploration: all-intra / IPPP fps versus batch size and kernel variant (not part of the bench contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen

def run(nframes, period, qp, steps=5, name="foremanlike"):
    base = clipgen.synth_clip(name, min(nframes, 300))
    clip = np.concatenate([base] * ((nframes + len(base) - 1) // len(base)))[:nframes]
    enc = capi.Encoder(352, 288, qp, qp, period, max_frames=nframes)
    enc.upload(clip)
    enc.encode_resident(0, nframes); enc.sync()
    enc.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        enc.encode_resident(0, nframes)
    enc.sync()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: round(v[0] / steps, 3) for k, v in enc.profile_get().items() if v[1]}
    enc.close()
    return nframes / dt, dt * 1e3, prof

if __name__ == "__main__":
    period = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    qp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    for n in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "30,300,600,1200,2400".split(","))]:
        fps, ms, prof = run(n, period, qp)
        print(f"n={n:5d} period={period} qp={qp}: {fps:10.0f} fps  {ms:8.3f} ms/step  {prof}", flush=True)
