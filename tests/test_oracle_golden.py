"""Pin the CPU oracle (oracle/icsp_oracle.c) against fixtures dumped from the compiled reference
(tools/make_golden.py).  Everything here is bit-exact: doubles are compared through their bit patterns."""
import glob
import hashlib
import json
import os
import re

import numpy as np
import pytest

from icspcodec_amd import clipgen
from oracle import pyoracle as po

W, H = 352, 288


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.int64)


@pytest.fixture(scope="module")
def blocks(golden_dir):
    return np.load(os.path.join(golden_dir, "blocks.npz"))


def test_costable_and_irt2(blocks):
    assert np.array_equal(bits(po.costable()), bits(blocks["costable"]))
    assert float(blocks["irt2"]) == po.irt2()
    assert po.irt2().hex() == "0x1.6a09e667f3bccp-1"      # SURVEY.md §8c probe


def test_dct_bit_exact(blocks):
    for e, d in zip(blocks["err"], blocks["dct"]):
        assert np.array_equal(bits(po.dct8x8(e)), bits(d))


def test_idct_bit_exact(blocks):
    for q, d in zip(blocks["iq"], blocks["idct"]):
        assert np.array_equal(bits(po.idct8x8(q)), bits(d))


def test_quantisers_luma_trunc_chroma_floor(blocks):
    qps = blocks["qps"]
    differ = 0
    for i, c in enumerate(blocks["coef"]):
        qd, qa = map(int, qps[i % len(qps)])
        for chroma, key in ((False, "luma"), (True, "chroma")):
            q, zz, iq, ac = po.quant(c, qd, qa, chroma)
            assert np.array_equal(q, blocks["q_" + key][i])
            assert np.array_equal(zz, blocks["zz_" + key][i])
            assert np.array_equal(iq, blocks["iq_" + key][i])
            assert ac == blocks["ac_" + key][i]
        differ += int(np.any(blocks["q_luma"][i] != blocks["q_chroma"][i]))
    assert differ > 0          # the luma/chroma rounding asymmetry is exercised


def test_padding_leaves_last_row_and_column_zero(blocks):
    for pad, key in ((16, "pad16"), (8, "pad8")):
        p = po.pad(blocks["plane"], pad)
        assert np.array_equal(p, blocks[key])
        assert not p[-1].any() and not p[:, -1].any()


def test_sad(blocks):
    for a, b, s in zip(blocks["sad_a"], blocks["sad_b"], blocks["sad"]):
        assert po.sad16(a, b) == s


def test_me_walks_reach_129_positions_in_4_states():
    pts = set()
    for s in range(4):
        dx, dy = po.me_walk(s)
        assert dx[0] == 0 and dy[0] == 0 and dx[1] == 0 and dy[1] == 0      # first two candidates coincide
        assert len(set(zip(dx.tolist(), dy.tolist()))) == 63
        pts |= set(zip(dx.tolist(), dy.tolist()))
        assert max(abs(dx).max(), abs(dy).max()) == 16
    assert len(pts) == 129
    dx, dy = po.me_walk(0)
    assert (dx.min(), dx.max(), dy.min(), dy.max()) == (-15, 16, -16, 15)


def test_me_static_pair_early_break_and_state(golden_dir):
    g = np.load(os.path.join(golden_dir, "me_static_pair.npz"))
    mx, my, ns = po.me_frame(g["cur"], g["prev"])
    assert np.array_equal(mx, g["mvx"]) and np.array_equal(my, g["mvy"])
    assert (ns < 64).any() and (ns == 64).any()           # both the break path and the full walk occur


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "frames_*.npz"))),
                         ids=lambda p: os.path.basename(p)[7:-4])
def test_frames_match_reference(path):
    m = re.match(r"frames_(\w+?)_(\d+)f_q(\d+)_(\d+)_p(\d+)\.npz", os.path.basename(path))
    name, n, qdc, qac, period = m.group(1), *map(int, m.groups()[1:])
    g = np.load(path)
    clip = clipgen.synth_clip(name, n)
    assert hashlib.sha256(clip.tobytes()).hexdigest() == str(g["clip_sha"])     # the generator is deterministic
    o = po.encode_sequence(clip, W, H, qdc, qac, period)
    for k in ("levels", "acflag", "mpm", "mvd"):
        assert np.array_equal(o[k], g[k]), k
    assert hashlib.sha256(o["recon"].tobytes()).hexdigest() == str(g["recon_sha"])
    assert np.array_equal(o["recon"][0], g["recon0"]) and np.array_equal(o["recon"][-1], g["recon_last"])
    # internal decisions: intra modes and raw motion vectors
    for f in range(n):
        if period == 0 or f % period == 0:
            r = po.intra_frame(clip[f], W, H, qdc, qac, want_dbg=True)
            assert np.array_equal(r["mode"], g["mode"][f])
        else:
            r = po.inter_frame(clip[f], o["recon"][f - 1], W, H, qdc, qac, want_dbg=True)
            assert np.array_equal(r["mv"], g["mv"][f])


def test_threaded_gop_queue_equals_single_thread():
    clip = clipgen.synth_clip("tablelike", 9)
    a = po.encode_sequence(clip, W, H, 16, 16, 3, nthreads=1)
    b = po.encode_sequence(clip, W, H, 16, 16, 3, nthreads=3)
    for k in a:
        assert np.array_equal(a[k], b[k])


def _streams(golden_dir):
    return json.load(open(os.path.join(golden_dir, "streams.json")))


def test_stream_level_recon_hashes_short(golden_dir):
    for s in _streams(golden_dir):
        if s["nframes"] > 12:
            continue
        clip = clipgen.synth_clip(s["clip"], s["nframes"])
        assert hashlib.sha256(clip.tobytes()).hexdigest() == s["clip_sha256"]
        period = s["intra_period"]
        o = po.encode_sequence(clip, W, H, s["qp"], s["qp"], period, nthreads=4)
        assert hashlib.sha256(o["recon"].tobytes()).hexdigest() == s["recon_sha256"], s


def test_stream_level_recon_hashes_baseline_configs(golden_dir):
    """BASELINE configs 1-3 at full length: foremanlike 300f all-intra QP16, stefanlike 300f period 10 QP8."""
    want = {("foremanlike", 300, 16, 0), ("stefanlike", 300, 8, 10)}
    seen = 0
    for s in _streams(golden_dir):
        if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) not in want:
            continue
        clip = clipgen.synth_clip(s["clip"], s["nframes"])
        o = po.encode_sequence(clip, W, H, s["qp"], s["qp"], s["intra_period"], nthreads=8)
        assert hashlib.sha256(o["recon"].tobytes()).hexdigest() == s["recon_sha256"], s
        seen += 1
    assert seen == 2
