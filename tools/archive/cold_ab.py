#!/usr/bin/env python3
"""Interleaved A/B of icsp_enc cold starts (300-frame CIF clip on tmpfs): cold_ab.py reps gap_s cfg1 cfg2 ...  (cfg: K=V,K=V or '-')"""
import json, os, subprocess, sys, tempfile, time, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from icspcodec_amd import clipgen
reps, gap = int(sys.argv[1]), float(sys.argv[2])
cfgs = [({} if a == "-" else dict(kv.split("=") for kv in a.split(","))) for a in sys.argv[3:]]
tmp = tempfile.mkdtemp(dir="/dev/shm")
name = clipgen.file_name("foremanlike", 300)
clipgen.synth_clip("foremanlike", 300).tofile(os.path.join(tmp, name))
res = {i: [] for i in range(len(cfgs))}
for rep in range(reps):
    for i, cfg in enumerate(cfgs):
        time.sleep(gap)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(R, "icspcodec_amd", "icsp_enc"), "-i", name, "-n", "300", "-q", "16", "--intraPeriod", "0", "--stats"],
                           cwd=tmp, env=dict(os.environ, **cfg), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        wall = time.perf_counter() - t0
        st = [l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")]
        d = json.loads(st[0][10:]) if st else {}
        res[i].append((wall, d.get("hip_start_s", -1), d.get("setup_worker0", {}).get("create_s", -1)))
for i, cfg in enumerate(cfgs):
    w = sorted(x[0] for x in res[i]); h = sorted(x[1] for x in res[i]); c = sorted(x[2] for x in res[i])
    fast = sum(1 for x in w if x < 0.2)
    print(f"gap {gap} {str(cfg):60s} wall median {statistics.median(w):.3f} min {w[0]:.3f} max {w[-1]:.3f}  fast(<0.2s) {fast}/{len(w)}  hip_start med {statistics.median(h):.3f}  create med {statistics.median(c):.3f}", flush=True)
