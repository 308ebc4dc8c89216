import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icspcodec_amd import capi, clipgen
import ctypes as C
W, H, n = 352, 288, 300
clip = clipgen.synth_clip("foremanlike", n)
enc = capi.Encoder(W, H, 16, 16, 0, max_frames=n)
src = capi.host_alloc_array(clip.shape, np.uint8); src[:] = clip
nb = enc.lib.icsp_bitstream_bound(C.byref(enc.params), n)
body = capi.host_alloc_array((nb,), np.uint8)
recon = capi.host_alloc_array((n, W*H*3//2), np.uint8)
for mode in ("norecon", "recon"):
    for rep in range(4):
        sys.stderr.write("---- %s rep %d\n" % (mode, rep)); sys.stderr.flush()
        t0 = time.perf_counter()
        enc.encode_packed(src, recon=recon if mode == "recon" else None, body=body)
        sys.stderr.write("---- call took %.1f us\n" % ((time.perf_counter() - t0) * 1e6)); sys.stderr.flush()
enc.close()
