"""The resident batches of bench.py's legs, so that the profiling tools (tools/leg_workload.py, tools/profile_round.sh) run the
very workloads the bench line quotes.  BASELINE.json configs[3] and configs[4]; the bundled clips are absent from the reference
checkout (/root/reference/.MISSING_LARGE_BLOBS), so everything is made by clipgen (synthetic, deterministic)."""
from __future__ import annotations

import numpy as np

from . import clipgen

W, H = 352, 288
P = W * H
CLIPS12 = ["akiyolike", "childrenlike", "coastguardlike", "containerlike", "footballlike", "foremanlike", "hallmonitorlike",
           "mobilelike", "motherdaughterlike", "newslike", "stefanlike", "tablelike"]
HD_W, HD_H, HD_PERIOD, HD_GOPS = 1920, 1088, 30, 100
HD_SRCS = ["foremanlike", "stefanlike", "mobilelike", "akiyolike"]


def clips12_units(period: int = 10):
    """(clip, first frame of the GOP, frames) for every closed GOP of the twelve clips: 339 units, 3390 frames."""
    units = []
    for name in CLIPS12:
        n = clipgen.CLIP_CLASSES[name]["nframes"]
        units += [(name, f, min(period, n - f)) for f in range(0, n, period)]
    return units


def clips12_shard(rank: int = 0, world: int = 1):
    """This rank's contiguous run of GOPs of configs[3] as one batch uint8[frames][W*H*3/2] and the units it holds."""
    units = clips12_units()
    per = len(units) // world
    lo = rank * per + min(rank, len(units) % world)
    mine = units[lo: lo + per + (1 if rank < len(units) % world else 0)]
    cache, parts = {}, []
    for name, f, cnt in mine:
        if name not in cache:
            cache = {name: clipgen.synth_clip(name)}                # whole clip once, GOPs are slices of it
        parts.append(cache[name][f: f + cnt])
    return np.concatenate(parts), mine, sum(u[2] for u in units)


def hd_gop(name: str) -> np.ndarray:
    """One 1920x1088 GOP of 30 frames: the CIF clip `name` tiled over the frame ("CIF-tiled macroblock grid", configs[4])."""
    c = clipgen.synth_clip(name, HD_PERIOD)
    fsz = HD_W * HD_H * 3 // 2
    out = np.empty((HD_PERIOD, fsz), np.uint8)
    for i in range(HD_PERIOD):
        y = c[i, :P].reshape(H, W)
        cb = c[i, P: P + P // 4].reshape(H // 2, W // 2)
        cr = c[i, P + P // 4:].reshape(H // 2, W // 2)
        out[i, : HD_W * HD_H] = np.tile(y, (4, 6))[:HD_H, :HD_W].ravel()
        out[i, HD_W * HD_H: HD_W * HD_H * 5 // 4] = np.tile(cb, (4, 6))[: HD_H // 2, : HD_W // 2].ravel()
        out[i, HD_W * HD_H * 5 // 4:] = np.tile(cr, (4, 6))[: HD_H // 2, : HD_W // 2].ravel()
    return out


def hd_shard(rank: int = 0, world: int = 1):
    """(first GOP, GOPs) of this rank and the distinct GOP contents it needs: GOP g shows clip HD_SRCS[g mod 4]."""
    per = HD_GOPS // world
    g_lo = rank * per + min(rank, HD_GOPS % world)
    g_n = per + (1 if rank < HD_GOPS % world else 0)
    gops = {nm: hd_gop(nm) for nm in {HD_SRCS[(g_lo + g) % 4] for g in range(g_n)}}
    return g_lo, g_n, gops
