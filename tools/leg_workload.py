#!/usr/bin/env python3
"""One loaded leg of bench.py as a workload for rocprofv3 (program directly after `--`):

    leg_workload.py config4 | config4_allintra | config5   [passes]

the same resident batch bench.py times (icspcodec_amd/workloads.py), `passes` passes (default 2) after one warm pass, the host
waiting after each.  Prints passes and frames, which tools/summarize_profiles.py needs to turn counter sums into per-pass figures."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, workloads
leg = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if leg in ("config4", "config4_allintra"):
    batch, mine, ntot = workloads.clips12_shard(0, 1)
    n = batch.shape[0]
    enc = capi.Encoder(workloads.W, workloads.H, 16, 16, 10 if leg == "config4" else 0, max_frames=n)
    enc.upload(batch)
elif leg == "config5":
    g_lo, g_n, gops = workloads.hd_shard(0, 1)
    L = workloads.HD_PERIOD
    n = g_n * L
    enc = capi.Encoder(workloads.HD_W, workloads.HD_H, 16, 16, L, max_frames=n)
    for g in range(g_n):
        enc.upload(gops[workloads.HD_SRCS[(g_lo + g) % 4]], first=g * L)
else:
    raise SystemExit("unknown leg " + leg)
enc.encode_resident(0, n)
enc.sync()
t0 = time.perf_counter()
for _ in range(passes):
    enc.encode_resident(0, n)
    enc.sync()
dt = (time.perf_counter() - t0) / passes
print(json.dumps({"leg": leg, "passes_total": passes + 1, "frames": n, "ms_per_pass_here": round(dt * 1e3, 3), "choice": enc.last_choice()}))
enc.close()
