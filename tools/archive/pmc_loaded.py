#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes on the loaded all-intra regime: pmc_loaded.py [nframes] (form from ICSP_INTRA_FORM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icspcodec_amd import capi, clipgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3390
base = clipgen.synth_clip("foremanlike", 300)
clip = np.concatenate([base] * ((n + 299) // 300))[:n]
enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=n)
enc.upload(clip)
for _ in range(3):
    enc.encode_resident(0, n)
    enc.sync()
enc.close()
