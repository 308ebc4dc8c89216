#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-call boundary (icsp_encode_gop / icsp_encode_gop_packed) by kind of caller memory:
gop_rate.py [period qp n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
period = int(sys.argv[1]) if len(sys.argv) > 1 else 0
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
W, H = 352, 288
nmb, fsz = 396, W * H * 3 // 2
clip = clipgen.synth_clip("stefanlike" if period else "foremanlike", n)
shapes = dict(levels=((n, nmb, 6, 64), np.int16), acflag=((n, nmb, 6), np.uint8), mpm=((n, nmb, 4), np.uint8), mvd=((n, nmb, 2), np.int8), recon=((n, fsz), np.uint8))
enc = capi.Encoder(W, H, qp, qp, period, max_frames=n)
for mem in ("pageable", "pinned"):
    if mem == "pinned":
        src = capi.host_alloc_array(clip.shape, np.uint8); src[:] = clip
        out = {k: capi.host_alloc_array(s, d) for k, (s, d) in shapes.items()}
    else:
        src = clip.copy(); out = {k: np.zeros(s, d) for k, (s, d) in shapes.items()}
    nb = enc.lib.icsp_bitstream_bound(__import__("ctypes").byref(enc.params), n); body = np.zeros(nb, np.uint8) if mem == "pageable" else capi.host_alloc_array((nb,), np.uint8)
    for _ in range(3):
        enc.encode(src, out=out)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); enc.encode(src, out=out); ts.append(time.perf_counter() - t0)
    tp = []
    for _ in range(3):
        enc.encode_packed(src, recon=out["recon"], body=body)
    for _ in range(7):
        t0 = time.perf_counter(); enc.encode_packed(src, recon=out["recon"], body=body); tp.append(time.perf_counter() - t0)
    print(f"period={period} n={n} {mem:9s}: encode_gop {n / sorted(ts)[3]:9.0f} fps ({sorted(ts)[3] * 1e3:.2f} ms; best {n / min(ts):.0f})   "
          f"encode_gop_packed (recon + bits) {n / sorted(tp)[3]:9.0f} fps ({sorted(tp)[3] * 1e3:.2f} ms)")
enc.close()
