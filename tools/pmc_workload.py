#!/usr/bin/env python3
"""The workload of the rocprofv3 --pmc passes: BASELINE configs[1] (300 CIF frames all-intra QP16) and configs[2] (300 frames,
--intraPeriod 10, QP8), three resident encode passes each (counters are per dispatch, kernels are serialised by the profiler).
The all-intra batch goes out as ONE launch of the luma kernel here (ICSP_I_GROUPS=1; the default is two launches on two streams,
which the profiler would serialise anyway): the counters of that launch are those of a bench step's launches together."""
import os, sys
os.environ["ICSP_I_GROUPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, clipgen
for name, q, period in (("foremanlike", 16, 0), ("stefanlike", 8, 10)):
    clip = clipgen.synth_clip(name, 300)
    enc = capi.Encoder(352, 288, q, q, period, max_frames=300)
    enc.upload(clip)
    for _ in range(3):
        enc.encode_resident(0, 300)
        enc.sync()
    enc.close()
