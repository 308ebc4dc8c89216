#!/usr/bin/env python3
"""icsp_enc on a 3000-frame clip (tmpfs) under environment variations: process wall, init, rate after set-up."""
import json, os, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from icspcodec_amd import clipgen
tmp = tempfile.mkdtemp(dir="/dev/shm")
name = "long_cif(352X288)_3000f.yuv"
c = clipgen.synth_clip("foremanlike", 300).tobytes()
with open(os.path.join(tmp, name), "wb") as f:
    for _ in range(10):
        f.write(c)
cfgs = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
for period in (0, 10):
    for cfg in cfgs:
        rows = []
        for rep in range(4):
            env = dict(os.environ, **cfg)
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(R, "icspcodec_amd", "icsp_enc"), "-i", name, "-n", "3000", "-q", "16", "--intraPeriod", str(period), "--stats"],
                               cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            wall = time.perf_counter() - t0
            st = [l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")]
            d = json.loads(st[0][10:]) if st else {}
            rows.append((wall, d.get("init_s", -1), d.get("encode_s", -1), d.get("e2e_fps_excl_init", -1), r.returncode))
        rows.sort()
        w, i, e, f, rc = rows[len(rows) // 2]
        print(f"period {period} {str(cfg):50s} median wall {w:.3f} s (min {rows[0][0]:.3f}) init {i:.3f} encode {e:.4f} fps_excl_init {f:.0f} (best {max(x[3] for x in rows):.0f}) rc {rc}", flush=True)
