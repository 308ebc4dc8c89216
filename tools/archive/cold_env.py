#!/usr/bin/env python3
"""Cold start of icsp_enc (300-frame CIF clip, tmpfs) under environment variations: process wall time and the program's own split."""
import json, os, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from icspcodec_amd import clipgen
tmp = tempfile.mkdtemp(dir="/dev/shm")
name = clipgen.file_name("foremanlike", 300)
clipgen.synth_clip("foremanlike", 300).tofile(os.path.join(tmp, name))
cfgs = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
for cfg in cfgs:
    rows = []
    for rep in range(5):
        env = dict(os.environ, **cfg)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(R, "icspcodec_amd", "icsp_enc"), "-i", name, "-n", "300", "-q", "16", "--intraPeriod", "0", "--stats"],
                           cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        wall = time.perf_counter() - t0
        st = [l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")]
        d = json.loads(st[0][10:]) if st else {}
        rows.append((wall, d.get("init_s", -1), d.get("hip_start_s", -1), d.get("setup_worker0", {}).get("create_s", -1), d.get("encode_s", -1), r.returncode))
    rows.sort()
    w, i, h, c, e, rc = rows[len(rows) // 2]
    print(f"{str(cfg):70s} median wall {w:.3f} s  (min {rows[0][0]:.3f})  init {i:.3f} hip_start {h:.3f} create {c:.3f} encode {e:.4f} rc {rc}", flush=True)
