"""Seeded fuzz: random geometries, quantisers, GOP lengths and pixel statistics (flat, extremes, noise, gradients, exact
repeats that trigger the search's early break) — every output of the HIP path against the oracle, the device bit packer
against the host writer, the device decoder against the decoder oracle."""
import os

import numpy as np
import pytest

from icspcodec_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
KEYS = ("levels", "acflag", "mpm", "mvd", "recon")


def _content(rng, kind, n, w, h):
    fsz = w * h * 3 // 2
    if kind == "noise":
        return rng.integers(0, 256, (n, fsz), dtype=np.uint8)
    if kind == "extremes":
        return (rng.integers(0, 2, (n, fsz), dtype=np.uint8) * 255).astype(np.uint8)
    if kind == "flat":
        return np.full((n, fsz), int(rng.integers(0, 256)), np.uint8)
    if kind == "gradient":
        base = (np.arange(fsz, dtype=np.int64) * int(rng.integers(1, 7)) // 3) % 256
        return np.stack([(base + 3 * f) % 256 for f in range(n)]).astype(np.uint8)
    if kind == "repeat":                       # identical frames: zero SADs, early breaks, state carry
        f0 = rng.integers(0, 256, fsz, dtype=np.uint8)
        return np.stack([f0] * n)
    # "shift": a smooth texture panning by a few pixels per frame, plus light noise
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for f in range(n):
        y = (np.sin((xx + 3 * f) / 7.0) * 60 + np.cos((yy - 2 * f) / 5.0) * 50 + 128 + rng.normal(0, 2, (h, w))).clip(0, 255)
        c = np.full((h // 2) * (w // 2) * 2, 128.0) + rng.normal(0, 6, (h // 2) * (w // 2) * 2)
        frames.append(np.concatenate([y.ravel(), c.clip(0, 255)]).astype(np.uint8))
    return np.stack(frames)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ICSP_FUZZ_SEEDS", "96"))))     # 1500 seeds were run once per round
def test_fuzz_case(seed):
    rng = np.random.default_rng(1000 + seed)
    w = 16 * int(rng.integers(2, 11)); h = 16 * int(rng.integers(1, 7))
    n = int(rng.integers(1, 8))
    period = int(rng.choice([0, 1, 2, 3, 4, 7]))
    qdc = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 255])); qac = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 255]))
    kind = ["noise", "extremes", "flat", "gradient", "repeat", "shift"][seed % 6]
    clip = _content(rng, kind, n, w, h)
    tag = f"seed={seed} {kind} {w}x{h} n={n} p={period} q={qdc}/{qac}: "
    enc = capi.Encoder(w, h, qdc, qac, period, max_frames=n)
    got = enc.encode(clip)
    bs = enc.pack_bitstream(0, n)
    enc.decode_resident(0, n)
    dec = enc.download(0, n, what=("recon",))["recon"]
    enc.close()
    want = po.encode_sequence(clip, w, h, qdc, qac, period)
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), tag + k
    assert bs == capi.write_bitstream(w, h, qdc, qac, period, want["levels"], want["acflag"], want["mpm"], want["mvd"]), tag + "bitstream"
    assert np.array_equal(dec, po.decode_sequence(want["levels"], want["mpm"], want["mvd"], w, h, qdc, qac, period)), tag + "decode"
