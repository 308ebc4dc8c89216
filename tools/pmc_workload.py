#!/usr/bin/env python3
"""The workload of the rocprofv3 --pmc passes (counters are per dispatch, kernels are serialised by the profiler):
  * BASELINE configs[1] the way bench.py runs it: two resident 300-frame all-intra batches (QP 16) encoded in turn, which makes the
    library place each batch whole on one stream and pick the luma kernel for 600 frames in flight (k_intra_luma8<3, ring>,
    one launch of 300 workgroups per batch) -- the dominant kernel of the bench line;
  * one 300-frame batch encoded again and again (ICSP_I_GROUPS=1: ONE launch of k_intra_luma32<8,4>, round 2's dominant kernel, kept
    for comparison);
  * configs[2]: 300 frames, --intraPeriod 10, QP 8, three passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icspcodec_amd import capi, clipgen
a = clipgen.synth_clip("foremanlike", 300)
b = clipgen.synth_clip("foremanlike", 300, first_frame=300)
enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=600)
enc.upload(a, first=0)
enc.upload(b, first=300)
for _ in range(3):
    enc.encode_resident(0, 300)
    enc.encode_resident(300, 300)
enc.sync()
enc.close()
os.environ["ICSP_I_GROUPS"] = "1"
for name, q, period in (("foremanlike", 16, 0), ("stefanlike", 8, 10)):
    clip = clipgen.synth_clip(name, 300)
    enc = capi.Encoder(352, 288, q, q, period, max_frames=300)
    enc.upload(clip)
    for _ in range(3):
        enc.encode_resident(0, 300)
        enc.sync()
    enc.close()
