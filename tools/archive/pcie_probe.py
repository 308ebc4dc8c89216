import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from icspcodec_amd import capi, clipgen
n = 300
c = clipgen.synth_clip("foremanlike", n)
enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=int(os.environ.get("MAXF", "900")))
shapes = dict(levels=((n, 396, 6, 64), np.int16), acflag=((n, 396, 6), np.uint8), mpm=((n, 396, 4), np.uint8), mvd=((n, 396, 2), np.int8), recon=((n, 352 * 288 * 3 // 2), np.uint8))
src = c.copy(); out = {k: np.zeros(sh, d) for k, (sh, d) in shapes.items()}
ts = []
for k in range(9):
    t0 = time.perf_counter(); enc.encode(src, out=out); ts.append(time.perf_counter() - t0)
print("pageable ms per call:", [round(t * 1e3, 2) for t in ts], "median fps", round(n / sorted(ts[2:])[3]))
enc.close()
