"""Where oracle/_ref has been built from /root/reference (this container), compare the restatement with the reference's
own object code on FRESH random inputs — beyond the committed fixtures.  Skipped on boxes without oracle/_ref."""
import numpy as np
import pytest

from icspcodec_amd import clipgen
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
W, H = 352, 288


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.int64)


def test_random_blocks_bit_exact():
    rng = np.random.default_rng()
    for _ in range(400):
        e = rng.integers(-255, 256, (8, 8))
        assert np.array_equal(bits(po.dct8x8(e)), bits(po.dct8x8(e, "ref")))
        q = (rng.integers(-2040, 2041, (8, 8)) * (rng.random((8, 8)) < rng.random())).astype(np.int32)
        assert np.array_equal(bits(po.idct8x8(q)), bits(po.idct8x8(q, "ref")))
        c = rng.normal(0, 300, (8, 8))
        for chroma in (False, True):
            qd, qa = int(rng.choice([1, 8, 16, 5])), int(rng.choice([1, 8, 16, 3]))
            a, b = po.quant(c, qd, qa, chroma), po.quant(c, qd, qa, chroma, "ref")
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]


@pytest.mark.parametrize("name,n,q,period", [("coastguardlike", 4, 16, 4), ("hallmonitorlike", 3, 8, 0), ("childrenlike", 5, 1, 5)])
def test_random_clip_sequences(name, n, q, period):
    first = int(np.random.default_rng().integers(0, 200))
    clip = clipgen.synth_clip(name, n, first_frame=first)
    o = po.encode_sequence(clip, W, H, q, q, period)
    r = po.ref_encode_frames(clip, W, H, q, q, period)
    for k in ("levels", "acflag", "mpm", "mvd", "recon"):
        assert np.array_equal(o[k], r[k]), (k, first)


@pytest.mark.parametrize("rep", range(6))
def test_random_geometry_and_quantisers(rep):
    """Any multiple of 16 and any quantiser pair: the reference's frame functions (not its CLI, which hard-codes CIF) against the
    restatement on fresh random shapes, contents and GOP lengths."""
    rng = np.random.default_rng()
    w, h = 16 * int(rng.integers(2, 30)), 16 * int(rng.integers(1, 20))
    n = int(rng.integers(1, 6))
    period = int(rng.choice([0, 1, 2, 3, 5]))
    qdc, qac = (int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 255])) for _ in range(2))
    kind = str(rng.choice(["noise", "extremes", "flat", "gradient", "repeat", "clip"]))
    seed = int(rng.integers(0, 1 << 20))
    clip = (clipgen.synth_clip("footballlike", n, width=w, height=h, first_frame=seed % 50) if kind == "clip"
            else clipgen.hashed_clip(kind, seed, n, w, h))
    o = po.encode_sequence(clip, w, h, qdc, qac, period)
    r = po.ref_encode_frames(clip, w, h, qdc, qac, period)
    for k in ("levels", "acflag", "mpm", "mvd", "recon"):
        assert np.array_equal(o[k], r[k]), (k, w, h, n, period, qdc, qac, kind, seed)


def test_me_with_exact_copies_and_state_carry():
    rng = np.random.default_rng()
    prev = rng.integers(0, 256, (H, W)).astype(np.uint8)
    cur = prev.copy()                                    # (0,0) is visited twice, so unchanged macroblocks break at step 1 ...
    cur[100:160, 50:200] = rng.integers(0, 256, (60, 150))   # ... and these ones then start in a non-default search state
    cur[200:260, 100:300] = np.roll(prev[200:260, 100:300], 2, axis=1)
    mo = po.me_frame(cur, prev)
    mr = po.me_frame(cur, prev, "ref")
    assert np.array_equal(mo[0], mr[0]) and np.array_equal(mo[1], mr[1])
    assert (mo[2] < 64).any()
