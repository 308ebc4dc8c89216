#!/usr/bin/env python3
"""Race hunt: re-encode / re-pack / re-decode the same resident batch many times and require identical bytes every time
(the kernels synchronise waves with raw s_barrier + LDS, streams with events; a race would show as a changing hash)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen

def run(name, n, q, period, reps, w=352, h=288):
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, q, q, period, max_frames=n)
    enc.upload(clip)
    ref = None
    for r in range(reps):
        enc.encode_resident(0, n)
        o = enc.download(0, n)
        bs = enc.pack_bitstream(0, n)
        enc.decode_resident(0, n)
        dec = enc.download(0, n, what=("recon",))["recon"]
        hsh = hashlib.sha256(b"".join(o[k].tobytes() for k in ("levels", "acflag", "mpm", "mvd", "recon")) + bs + dec.tobytes()).hexdigest()
        if ref is None:
            ref = hsh
        elif hsh != ref:
            print(f"MISMATCH {name} n={n} q={q} p={period} rep={r}")
            return False
    enc.close()
    print(f"ok {name} {w}x{h} n={n} q={q} p={period} x{reps}: {ref[:16]}")
    return True

ok = True
ok &= run("foremanlike", 300, 16, 0, 25)
ok &= run("stefanlike", 300, 8, 10, 25)
ok &= run("staticlike", 300, 1, 6, 40)      # every frame flagged: the fused kernel's last-arriver hand-off on every P step
ok &= run("staticlike", 60, 16, 3, 40)
ok &= run("mobilelike", 64, 16, 4, 10, 704, 576)
ok &= run("tablelike", 6, 16, 3, 5, 1920, 1088)
sys.exit(0 if ok else 1)
