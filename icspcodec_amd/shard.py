"""Closed-GOP sharding across the GPUs of one node — the multi-GPU analogue of the reference's GOP job queue
(ICSP_thread.cpp:39-77, ICSP_Codec_Encoder_source.cpp:186-213).  One process per GPU; shards are independent (a P frame
only references the previous frame of its own GOP), so there is NO data-path collective: the only communication is the
host-side gather of per-GOP outputs in frame order for the sequential bit packer, and the max-over-ranks of the timing.
"""
from __future__ import annotations

import os

import numpy as np


def gop_shards(nframes: int, intra_period: int, world: int):
    """Contiguous runs of whole GOPs per rank: [(first_frame, n_frames)] * world (ranks beyond the GOP count get
    (nframes, 0)).  Same split as the C++ host (icsp_enc_main.cpp)."""
    L = intra_period if intra_period > 0 else 1
    ngop = (nframes + L - 1) // L
    out, g0 = [], 0
    for r in range(world):
        gcount = ngop // world + (1 if r < ngop % world else 0)
        first = min(g0 * L, nframes)
        last = min(nframes, (g0 + gcount) * L)
        out.append((first, last - first))
        g0 += gcount
    return out


def init_distributed():
    """(rank, world, local_rank, dist-or-None).  Backend: nccl (= RCCL) when a GPU is present, else gloo."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local, None
    import torch
    import torch.distributed as dist
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    return rank, world, local, dist


def max_over_ranks(value: float, dist) -> float:
    if dist is None:
        return value
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_in_frame_order(local: np.ndarray, shards, rank: int, dist):
    """Rank 0 receives every rank's per-frame array and returns the concatenation in frame order (others: None)."""
    if dist is None:
        return local
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    world = len(shards)
    per = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    nmax = max(c for _, c in shards)
    buf = np.zeros((nmax,) + local.shape[1:], dtype=local.dtype)
    buf[: local.shape[0]] = local
    t = torch.from_numpy(buf.view(np.uint8).reshape(-1)).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    if dev == "cuda":
        # NCCL/RCCL has no gather to a list on every version: all_gather is fine at these sizes
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
    else:
        dist.gather(t, outs, dst=0)
    if rank != 0:
        return None
    parts = []
    for r, (first, cnt) in enumerate(shards):
        a = outs[r].cpu().numpy().view(local.dtype).reshape((nmax,) + local.shape[1:])
        parts.append(a[:cnt])
    return np.concatenate(parts, axis=0)
