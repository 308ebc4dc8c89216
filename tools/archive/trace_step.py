#!/usr/bin/env python3
"""Kernel timeline of ONE resident encode, for `rocprofv3 --kernel-trace`:   rocprofv3 --kernel-trace -d DIR -- python3 tools/trace_step.py [period qp nframes name reps]
(then tools/trace_table.py DIR prints start/duration/gap of the last pass's kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen

period = int(sys.argv[1]) if len(sys.argv) > 1 else 10
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
name = sys.argv[4] if len(sys.argv) > 4 else "stefanlike"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 60
base = clipgen.synth_clip(name, min(n, 300))
clip = np.concatenate([base] * ((n + len(base) - 1) // len(base)))[:n]
enc = capi.Encoder(352, 288, qp, qp, period, max_frames=n)
enc.upload(clip)
for _ in range(reps):
    enc.encode_resident(0, n)
    enc.sync()
enc.close()
