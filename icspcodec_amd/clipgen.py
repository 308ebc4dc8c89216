"""Deterministic synthetic I420 clips standing in for the reference's bundled CIF sequences.

The reference ships its twelve ``data/*_cif(352X288)_*.yuv`` clips as large blobs that are absent
from the checkout (``/root/reference/.MISSING_LARGE_BLOBS``), so every BASELINE configuration is
restated on clips produced here.  Everything is integer arithmetic on a
counter-based hash (splitmix64 finaliser), so the same bytes come out on any machine, any numpy
version, and the golden fixtures made from the compiled reference stay valid on the GPU box.

Layout of one clip: ``uint8[nframes, W*H*3/2]`` — per frame the Y plane, then Cb, then Cr, exactly
the planar I420 order the reference loader reads (ICSP_Codec_Encoder_source.cpp:274-279).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _hash2d(seed: int, salt: int, h: int, w: int) -> np.ndarray:
    """uint64[h, w] of hashed (seed, salt, y, x)."""
    idx = (np.arange(h, dtype=np.uint64)[:, None] << np.uint64(20)) | np.arange(w, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x1000003 + salt * 0x10001 + 0x5bd1e995) & 0xFFFFFFFFFFFFFFFF)
        return _mix(idx + base * np.uint64(0x2545F4914F6CDD1D))


def _box_blur_wrap(a: np.ndarray, r: int) -> np.ndarray:
    """Integer box blur with toroidal wrap; a is int64[h, w]; returns floor(mean)."""
    if r <= 0:
        return a
    acc = np.zeros_like(a)
    for d in range(-r, r + 1):
        acc += np.roll(a, d, axis=0)
    acc2 = np.zeros_like(a)
    for d in range(-r, r + 1):
        acc2 += np.roll(acc, d, axis=1)
    k = (2 * r + 1) ** 2
    return acc2 // k


def _texture(seed: int, salt: int, size: int, blur: int, lo: int, hi: int) -> np.ndarray:
    """Band-limited periodic texture int64[size, size] stretched to [lo, hi]."""
    n = (_hash2d(seed, salt, size, size) >> np.uint64(40)).astype(np.int64) & 0xFFFF
    b = _box_blur_wrap(n, blur)
    mn, mx = int(b.min()), int(b.max())
    if mx == mn:
        return np.full((size, size), (lo + hi) // 2, dtype=np.int64)
    return lo + (b - mn) * (hi - lo) // (mx - mn)


# (pan_dx, pan_dy, blur radius, contrast lo, hi, noise amplitude, number of moving rectangles, static)
CLIP_CLASSES = {
    "akiyolike":          dict(seed=3,  pan=(0, 0), blur=6, lo=40,  hi=200, noise=0, rects=1, nframes=300),
    "childrenlike":       dict(seed=4,  pan=(1, 1), blur=3, lo=20,  hi=235, noise=2, rects=3, nframes=300),
    "coastguardlike":     dict(seed=5,  pan=(2, 0), blur=2, lo=30,  hi=220, noise=2, rects=2, nframes=300),
    "containerlike":      dict(seed=6,  pan=(1, 0), blur=5, lo=50,  hi=210, noise=1, rects=1, nframes=300),
    "footballlike":       dict(seed=7,  pan=(5, 3), blur=2, lo=10,  hi=245, noise=3, rects=4, nframes=90),
    "foremanlike":        dict(seed=1,  pan=(1, 0), blur=3, lo=30,  hi=225, noise=2, rects=2, nframes=300),
    "hallmonitorlike":    dict(seed=8,  pan=(0, 0), blur=4, lo=35,  hi=215, noise=1, rects=2, nframes=300),
    "mobilelike":         dict(seed=9,  pan=(1, 2), blur=1, lo=5,   hi=250, noise=2, rects=3, nframes=300),
    "motherdaughterlike": dict(seed=10, pan=(0, 0), blur=6, lo=45,  hi=190, noise=1, rects=1, nframes=300),
    "newslike":           dict(seed=11, pan=(0, 0), blur=4, lo=30,  hi=220, noise=0, rects=2, nframes=300),
    "stefanlike":         dict(seed=2,  pan=(6, 1), blur=1, lo=5,   hi=250, noise=2, rects=2, nframes=300),
    "tablelike":          dict(seed=12, pan=(3, 1), blur=3, lo=25,  hi=230, noise=2, rects=3, nframes=300),
    # identical frames with exactly-flat regions: exercises the SAD==0 early break and the carried
    # search-direction state of motionEstimation (ICSP_Codec_Encoder_source.cpp:2095, 2136-2141)
    "staticlike":         dict(seed=13, pan=(0, 0), blur=5, lo=60,  hi=180, noise=0, rects=0, nframes=4, flat=True),
}


def file_name(name: str, nframes: int, width: int = 352, height: int = 288) -> str:
    """Reference-shaped file name: the output prefix is cut at the first '_' (encoder_main.cpp:13)."""
    tag = "cif" if (width, height) == (352, 288) else "hd"
    return f"{name}_{tag}({width}X{height})_{nframes}f.yuv"


def synth_clip(name: str = "foremanlike", nframes: int | None = None, width: int = 352, height: int = 288,
               first_frame: int = 0) -> np.ndarray:
    """Return uint8[nframes, width*height*3//2] for clip class ``name`` (frames first_frame ...)."""
    p = CLIP_CLASSES[name]
    if nframes is None:
        nframes = p["nframes"]
    seed = p["seed"]
    size = 512
    while size < max(width, height):
        size *= 2
    ty = _texture(seed, 1, size, p["blur"], p["lo"], p["hi"])
    tcb = _texture(seed, 2, size // 2, p["blur"] + 4, 90, 170)
    tcr = _texture(seed, 3, size // 2, p["blur"] + 4, 80, 180)
    if p.get("flat"):
        # carve exactly-flat plateaus into the texture (values that survive coarse quantisation)
        ty = ty.copy()
        ty[:, : size // 3] = 128
        ty[size // 4: size // 2, :] = 96
        tcb = tcb.copy(); tcr = tcr.copy()
        tcb[:, : size // 6] = 128
        tcr[:, : size // 6] = 128
    dx, dy = p["pan"]
    nrect = p["rects"]
    cw, ch = width // 2, height // 2
    out = np.empty((nframes, width * height * 3 // 2), dtype=np.uint8)
    ys = np.arange(height)[:, None]
    xs = np.arange(width)[None, :]
    cys = np.arange(ch)[:, None]
    cxs = np.arange(cw)[None, :]
    # rectangle descriptors from the hash: size, start, velocity, brightness
    rh = _mix(np.arange(64, dtype=np.uint64) + np.uint64(seed * 7919))
    for i in range(nframes):
        n = first_frame + i
        ox, oy = n * dx, n * dy
        Y = ty[(ys + oy) % size, (xs + ox) % size].copy()
        Cb = tcb[(cys + oy // 2) % (size // 2), (cxs + ox // 2) % (size // 2)].copy()
        Cr = tcr[(cys + oy // 2) % (size // 2), (cxs + ox // 2) % (size // 2)].copy()
        for r in range(nrect):
            w_r = 24 + int(rh[4 * r] % np.uint64(72))
            h_r = 24 + int(rh[4 * r + 1] % np.uint64(56))
            vx = int(rh[4 * r + 2] % np.uint64(9)) - 4
            vy = int(rh[4 * r + 3] % np.uint64(5)) - 2
            x0 = (int(rh[4 * r + 2] >> np.uint64(20)) % width + n * vx) % width
            y0 = (int(rh[4 * r + 3] >> np.uint64(20)) % height + n * vy) % height
            val = 40 + int(rh[4 * r] >> np.uint64(24)) % 180
            x1, y1 = min(width, x0 + w_r), min(height, y0 + h_r)
            # object keeps its own (non-panning) texture so the motion field is not uniform
            obj = ty[(ys[y0:y1] * 3) % size, (xs[:, x0:x1] * 3 + 17 * r) % size]
            Y[y0:y1, x0:x1] = (obj + val) // 2
            Cb[y0 // 2:y1 // 2, x0 // 2:x1 // 2] = 100 + 10 * r
            Cr[y0 // 2:y1 // 2, x0 // 2:x1 // 2] = 150 - 12 * r
        amp = p["noise"]
        if amp:
            ny = (_hash2d(seed, 100 + n, height, width) >> np.uint64(33)).astype(np.int64) % (2 * amp + 1) - amp
            Y = Y + ny
            nc = (_hash2d(seed, 5000 + n, ch, cw) >> np.uint64(33)).astype(np.int64) % 3 - 1
            Cb = Cb + nc
            Cr = Cr - nc
        frame = out[i]
        frame[: width * height] = np.clip(Y, 0, 255).astype(np.uint8).ravel()
        frame[width * height: width * height + cw * ch] = np.clip(Cb, 0, 255).astype(np.uint8).ravel()
        frame[width * height + cw * ch:] = np.clip(Cr, 0, 255).astype(np.uint8).ravel()
    return out


def write_clip(path: str, clip: np.ndarray) -> None:
    clip.tofile(path)


def psnr_y(orig: np.ndarray, recon: np.ndarray, width: int, height: int) -> float:
    """Mean over frames of 20*log10(255/sqrt(MSE_Y)) — the decoder's definition
    (ICSP_Codec_Decoder.h:331-348), luma only."""
    n = orig.shape[0]
    o = orig[:, : width * height].astype(np.float64)
    r = recon[:, : width * height].astype(np.float64)
    mse = ((o - r) ** 2).mean(axis=1)
    mse = np.maximum(mse, 1e-12)
    return float((20.0 * np.log10(255.0 / np.sqrt(mse))).mean())


def hashed_clip(kind: str, seed: int, nframes: int, width: int, height: int) -> np.ndarray:
    """Deterministic stress content for any geometry (integer hash only, like synth_clip): ``noise`` (uniform bytes),
    ``extremes`` (every byte 0 or 255), ``flat`` (one value), ``gradient`` (ramps), ``repeat`` (identical noisy frames:
    zero SADs, the early break and the carried search state of motionEstimation,
    ICSP_Codec_Encoder_source.cpp:2095, 2136-2141).  Used by the geometry fixtures (tools/make_golden.py), whose expected
    outputs come from the compiled reference, so the bytes must be the same on every box."""
    fsz = width * height * 3 // 2
    rows = (fsz + 1023) // 1024

    def noise(salt):
        return ((_hash2d(seed, salt, rows, 1024) >> np.uint64(29)) & np.uint64(0xFF)).astype(np.uint8).ravel()[:fsz]
    if kind == "noise":
        return np.stack([noise(f) for f in range(nframes)])
    if kind == "extremes":
        return np.stack([np.where(noise(f) & 1, 255, 0).astype(np.uint8) for f in range(nframes)])
    if kind == "flat":
        return np.full((nframes, fsz), (seed * 37 + 11) % 256, np.uint8)
    if kind == "gradient":
        base = np.arange(fsz, dtype=np.int64) * (1 + seed % 6) // 3
        return np.stack([((base + 3 * f) % 256).astype(np.uint8) for f in range(nframes)])
    if kind == "repeat":
        return np.repeat(noise(0)[None, :], nframes, axis=0)
    raise ValueError(kind)
