"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the GOP-sharding host logic.  Each rank encodes its shard
with the oracle (tests may use it), rank 0 gathers in frame order, packs the bitstream with the product's host packer
and compares with the single-process result."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icspcodec_amd import capi, clipgen, shard  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

W, H = 352, 288
name, nframes, qp, period, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rank, world, local, dist = shard.init_distributed()
shards = shard.gop_shards(nframes, period, world)
first, cnt = shards[rank]
clip = clipgen.synth_clip(name, cnt, first_frame=first) if cnt else np.zeros((0, W * H * 3 // 2), np.uint8)
t0 = time.perf_counter()
o = po.encode_sequence(clip, W, H, qp, qp, period) if cnt else po._alloc(0, W, H)
dt = shard.max_over_ranks(time.perf_counter() - t0, dist)
full = {k: shard.gather_in_frame_order(v, shards, rank, dist) for k, v in o.items()}
if rank == 0:
    bs = capi.write_bitstream(W, H, qp, qp, period, full["levels"], full["acflag"], full["mpm"], full["mvd"])
    json.dump({"world": world, "shards": shards, "max_dt": dt, "bin_sha256": hashlib.sha256(bs).hexdigest(),
               "bin_bytes": len(bs), "recon_sha256": hashlib.sha256(full["recon"].tobytes()).hexdigest()}, open(out, "w"))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
