#!/usr/bin/env python3
"""What, done earlier in a process, costs a later IPPP context its speed?  exp_prelude.py <stages>  (letters: a = an all-intra context
used and closed; g = icsp_encode_gop from plain and pinned memory on it; p = pack_bits; d = decode; m = encode_resident_many)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
st = sys.argv[1] if len(sys.argv) > 1 else ""
n = 300
if st:
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=2 * n)
    c = clipgen.synth_clip("foremanlike", n)
    enc.upload(c, first=0); enc.upload(c, first=n)
    for k in range(40):
        enc.encode_resident((k & 1) * n, n)
    enc.sync()
    if "m" in st:
        for k in range(40):
            enc.encode_resident_many([[(0, 150), (300, 150)], [(150, 150), (450, 150)]][k & 1])
        enc.sync()
    if "g" in st:
        src = c.copy()
        for _ in range(3): enc.encode(src)
        srcp = capi.host_alloc_array(c.shape, np.uint8); srcp[:] = c
        for _ in range(3): enc.encode(srcp)
        capi.host_free_array(srcp)
    if "c" in st:
        enc.lib.icsp_copy_streams(enc.ctx, 1)
    if "u" in st:                       # uploads / downloads through the shared transfer streams
        enc.lib.icsp_copy_streams(enc.ctx, 1)
        for _ in range(3):
            enc.upload(c, first=0)
            enc.download(0, n, what=("recon",))
    if "p" in st:
        hb = np.empty(64 << 20, np.uint8)
        enc.upload(c); enc.encode_resident(0, n)
        for _ in range(3): enc.pack_bits(0, n, hb)
    if "d" in st:
        for _ in range(4): enc.decode_resident(0, n)
        enc.sync()
    enc.close()
    if "w" in st:
        time.sleep(2.0)
if "S" in st or "T" in st or "K" in st:           # two plain streams made directly with the runtime: S = never used, T = one copy each, K = one kernel-side memset each
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hs = [C.c_void_p(), C.c_void_p()]
    for h in hs:
        assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
    if "T" in st or "K" in st:
        dv = C.c_void_p(); hv = C.c_void_p()
        assert hip.hipMalloc(C.byref(dv), 1 << 20) == 0 and hip.hipHostMalloc(C.byref(hv), 1 << 20, 0) == 0
        if "T" in st:
            assert hip.hipMemcpyAsync(dv, hv, 1 << 20, 1, hs[0]) == 0          # H2D
            assert hip.hipMemcpyAsync(hv, dv, 1 << 20, 2, hs[1]) == 0          # D2H
        else:
            assert hip.hipMemsetAsync(dv, 0, 1 << 20, hs[0]) == 0 and hip.hipMemsetAsync(dv, 0, 1 << 19, hs[1]) == 0
        for h in hs:
            hip.hipStreamSynchronize(h)
enc = capi.Encoder(352, 288, 8, 8, 10, max_frames=2 * n)
for r in range(2):
    enc.upload(clipgen.synth_clip("stefanlike", n, first_frame=r * n), first=r * n)
for k in range(110):
    enc.encode_resident((k & 1) * n, n)
enc.sync()
if "L" in st:                           # two plain streams made AFTER this context's four
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hl = [C.c_void_p(), C.c_void_p()]
    for h in hl:
        assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
    for k in range(20):
        enc.encode_resident((k & 1) * n, n)
    enc.sync()
dts = []
for rep in range(5):
    t0 = time.perf_counter()
    for k in range(50):
        enc.encode_resident((k & 1) * n, n)
    enc.sync()
    dts.append(time.perf_counter() - t0)
dts.sort()
print(f"prelude '{st}': IPPP two ranges alternating, 50-step regions: median {50 * n / dts[2]:.0f} fps")
enc.close()
