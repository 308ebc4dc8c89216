#!/usr/bin/env python3
"""Experiment: does an initialised RCCL communicator (its streams) disturb the two-stream P-step overlap?  Single rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from icspcodec_amd import capi, clipgen

def ippp():
    clip = clipgen.synth_clip("stefanlike", 300)
    enc = capi.Encoder(352, 288, 8, 8, 10, max_frames=300)
    enc.upload(clip)
    for _ in range(3): enc.encode_resident(0, 300)
    enc.sync()
    t0 = time.perf_counter()
    for _ in range(10): enc.encode_resident(0, 300)
    enc.sync()
    dt = (time.perf_counter() - t0) / 10
    enc.close()
    return 300 / dt

torch.cuda.set_device(0)
print("before rccl:", round(ippp()))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("after rccl init + collective:", round(ippp()))
x = torch.randn(1024, 1024, device="cuda"); s = torch.cuda.Stream(); 
with torch.cuda.stream(s): y = x @ x
torch.cuda.synchronize()
print("after extra torch stream:", round(ippp()))
dist.destroy_process_group()
