"""Parity at the shapes round 1 left out (VERDICT r01 "What's weak" 1-3):
  * BASELINE configs[4] at its own shape: 1920x1088, --intraPeriod 30, two full GOPs (29 dependent P steps each), encode +
    device bit packer + device decode;
  * a CIF period-30 GOP of static content (every P frame raises its early-break flag, so the four-state search and the
    state carry of motionEstimation, ENC:2095 vs 2106-2141, run on 29 consecutive P steps);
  * the two-launch form of the P step (k_me<true,16> + k_frame_serial) with raised flags, at CIF through ICSP_NO_FUSE=1 and
    at 1088p (where it is the only form);
  * the fused form (k_serial_fused, last-arriver hand-off) with EVERY frame flagged in grids far larger than the chip can
    hold at once, and with four contexts sharing the device.
ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp."""
import os
import threading

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
W, H = 352, 288
KEYS = ("levels", "acflag", "mpm", "mvd", "recon")
NT = min(os.cpu_count() or 1, 64)


def _cmp(got, want, ctx=""):
    for k in KEYS:
        if not np.array_equal(got[k], want[k]):
            bad = np.argwhere(got[k] != want[k])
            raise AssertionError(f"{ctx}{k}: {len(bad)} mismatches, first at {bad[0].tolist()}: "
                                 f"got {got[k][tuple(bad[0])]} want {want[k][tuple(bad[0])]}")


def _static_clip(n, w=W, h=H):
    """n identical frames with exactly flat regions: every macroblock of a flat region finds SAD == 0 twice."""
    return np.repeat(clipgen.synth_clip("staticlike", 1, width=w, height=h), n, axis=0)


def _full_check(clip, w, h, q, period, ctx):
    n = clip.shape[0]
    enc = capi.Encoder(w, h, q, q, period, max_frames=n)
    got = enc.encode(clip)
    bs = enc.pack_bitstream(0, n)
    enc.decode_resident(0, n)
    dec = enc.download(0, n, what=("recon",))["recon"]
    enc.close()
    want = po.encode_sequence(clip, w, h, q, q, period, nthreads=NT)
    _cmp(got, want, ctx)
    assert bs == capi.write_bitstream(w, h, q, q, period, want["levels"], want["acflag"], want["mpm"], want["mvd"]), ctx
    assert np.array_equal(dec, po.decode_sequence(want["levels"], want["mpm"], want["mvd"], w, h, q, q, period)), ctx
    return want


def test_config5_1088p_period30_two_gops():
    """BASELINE configs[4]: 1920x1088 (1080 is not a multiple of 16, ENC:322), --intraPeriod 30: 60 frames = 2 GOPs."""
    w, h = 1920, 1088
    clip = clipgen.synth_clip("tablelike", 60, width=w, height=h)
    _full_check(clip, w, h, 16, 30, "1088p p30: ")


def test_config5_full_3000_frames_100_gops():
    """BASELINE configs[4] at its FULL size: 3000 frames of 1920x1088 = 100 closed GOPs of 30 resident at once (40 GB
    of HBM), one pass -- the batch bench.py's `config5` leg times.  GOP g shows content (g mod 4) (icspcodec_amd/workloads.py: the CIF
    clips tiled over the frame): every one of the 100 GOPs must reproduce the oracle's levels-and-all result of its content, i.e. 25
    GOPs per content, bit for bit in the reconstruction, and the first GOP of each content in every output array."""
    from icspcodec_amd import workloads
    w, h, L, ngop = workloads.HD_W, workloads.HD_H, workloads.HD_PERIOD, workloads.HD_GOPS
    g_lo, g_n, gops = workloads.hd_shard(0, 1)
    assert (g_lo, g_n) == (0, ngop)
    srcs = workloads.HD_SRCS
    allw = po.encode_sequence(np.concatenate([gops[nm] for nm in srcs]), w, h, 16, 16, L, nthreads=min(4, NT))     # one GOP per thread
    want = {nm: {k: v[j * L: (j + 1) * L] for k, v in allw.items()} for j, nm in enumerate(srcs)}
    enc = capi.Encoder(w, h, 16, 16, L, max_frames=ngop * L)
    for g in range(ngop):
        enc.upload(gops[srcs[g % 4]], first=g * L)
    enc.encode_resident(0, ngop * L)
    enc.sync()
    for g in range(ngop):
        nm = srcs[g % 4]
        if g < 4:
            _cmp(enc.download(g * L, L), want[nm], f"GOP {g} ({nm}): ")
        else:
            rec = enc.download(g * L, L, what=("recon",))["recon"]
            assert np.array_equal(rec, want[nm]["recon"]), f"GOP {g} ({nm}): reconstruction differs from the oracle"
    enc.close()


def test_cif_period30_static_content():
    """29 consecutive flagged P steps per GOP (two GOPs + a ragged third)."""
    clip = _static_clip(67)
    want = _full_check(clip, W, H, 16, 30, "CIF static p30: ")
    assert (want["mvd"][1:30] == 0).mean() > 0.5        # static content: mostly zero vectors ... and the flags were up:
    mx, my, ns = po.me_frame(clip[1, :W * H].reshape(H, W), want["recon"][0, :W * H].reshape(H, W))
    assert (ns < 64).any()


def test_cif_period30_moving_content():
    clip = clipgen.synth_clip("stefanlike", 60)
    _full_check(clip, W, H, 8, 30, "CIF stefan p30: ")


@pytest.fixture
def no_fuse():
    os.environ["ICSP_NO_FUSE"] = "1"              # read by icsp_create
    yield
    del os.environ["ICSP_NO_FUSE"]


def test_two_launch_form_cif_flagged(no_fuse):
    """k_me<true,16> with need == true + k_frame_serial at CIF (ICSP_NO_FUSE=1), static and moving content."""
    _full_check(_static_clip(24), W, H, 16, 6, "no-fuse static: ")
    _full_check(_static_clip(8), W, H, 1, 4, "no-fuse static q1: ")
    clip = clipgen.synth_clip("akiyolike", 12)
    _full_check(clip, W, H, 16, 4, "no-fuse akiyo: ")


def test_two_launch_form_1088p_flagged():
    """nmb >= 2048 always takes the two-launch form; a static 1088p clip raises the flag on every P frame."""
    w, h = 1920, 1088
    clip = _static_clip(6, w, h)
    want = _full_check(clip, w, h, 16, 3, "1088p static: ")
    mx, my, ns = po.me_frame(clip[1, :w * h].reshape(h, w), want["recon"][0, :w * h].reshape(h, w))
    assert (ns < 64).any()


def _hash_rows(a):
    return a.reshape(a.shape[0], -1).astype(np.uint64).sum(axis=1)


def test_fused_form_every_frame_flagged_beyond_residency():
    """1200 CIF GOPs of static content in ONE call: 1200 serial workgroups + 118 800 search workgroups per launch, every
    frame flagged -- far more than the 1024 workgroups of this kernel the chip holds.  The old spin-wait could be starved
    here; the last-arriver hand-off cannot.  One GOP's oracle output is the expected output of all."""
    n, period = 2400, 2
    clip = _static_clip(n)
    enc = capi.Encoder(W, H, 16, 16, period, max_frames=n)
    enc.upload(clip)
    enc.encode_resident(0, n)
    want = po.encode_sequence(clip[:2], W, H, 16, 16, period)
    for f0 in range(0, n, 400):
        got = enc.download(f0, 400)
        for k in KEYS:
            g = got[k].reshape((200, 2) + got[k].shape[1:])
            assert (g == want[k][None]).all(), (k, f0)
    enc.close()


def test_fused_form_4cif_flagged():
    w, h, n, period = 704, 576, 640, 2
    clip = _static_clip(n, w, h)
    enc = capi.Encoder(w, h, 16, 16, period, max_frames=n)
    enc.upload(clip)
    enc.encode_resident(0, n)
    want = po.encode_sequence(clip[:2], w, h, 16, 16, period)
    for f0 in range(0, n, 160):
        got = enc.download(f0, 160)
        for k in KEYS:
            g = got[k].reshape((80, 2) + got[k].shape[1:])
            assert (g == want[k][None]).all(), (k, f0)
    enc.close()


def test_four_contexts_share_one_device_flagged():
    """icsp_enc --EnMultiThread 4 on a one-GPU box puts four contexts on one device (icsp_enc_main.cpp): four host threads,
    300 flagged GOPs each, all launching at once."""
    n, period = 600, 2
    clip = _static_clip(n)
    want = po.encode_sequence(clip[:2], W, H, 16, 16, period)
    errs = []

    def work(i):
        try:
            enc = capi.Encoder(W, H, 16, 16, period, max_frames=n)
            enc.upload(clip)
            for _ in range(3):
                enc.encode_resident(0, n)
            got = enc.download(0, n)
            enc.close()
            for k in KEYS:
                g = got[k].reshape((n // 2, 2) + got[k].shape[1:])
                if not (g == want[k][None]).all():
                    errs.append((i, k))
        except Exception as e:           # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("name,n,q,period,groups", [("stefanlike", 83, 8, 10, 2), ("foremanlike", 41, 16, 10, 2), ("stefanlike", 40, 8, 10, 1),
                                                    ("foremanlike", 7, 16, 10, 2), ("foremanlike", 1, 16, 10, 2),
                                                    ("staticlike", 24, 16, 2, 2), ("stefanlike", 33, 8, 3, 2)])
def test_back_to_back_passes_of_one_range_overlap(name, n, q, period, groups, monkeypatch):
    """encode_range called again and again on the same resident range without a sync in between: the I frames of pass N+1 run
    beside P steps 2.. of pass N (icsp_sched.cpp, encode_range).  Ragged ends (a last GOP that stops before, at, or after its
    first P frame; a lone I frame), one and two GOP groups; then a different range, an upload and a last pass."""
    monkeypatch.setenv("ICSP_P_GROUPS", str(groups))
    clip = clipgen.synth_clip(name, n)
    want = po.encode_sequence(clip, W, H, q, q, period)
    enc = capi.Encoder(W, H, q, q, period, max_frames=n)
    enc.upload(clip)
    for _ in range(6):
        enc.encode_resident(0, n)
    _cmp(enc.download(0, n), want, "six passes")
    if n > period:                                       # another range in between, then the whole again, twice
        enc.encode_resident(period, n - period)
        enc.encode_resident(0, n)
        enc.encode_resident(0, n)
        _cmp(enc.download(0, n), want, "after a sub-range")
    other = clipgen.synth_clip("mobilelike" if name != "mobilelike" else "foremanlike", n)
    enc.upload(other)                                    # new input: the next pass must not start before the upload has landed
    enc.encode_resident(0, n)
    enc.encode_resident(0, n)
    _cmp(enc.download(0, n), po.encode_sequence(other, W, H, q, q, period), "after an upload")
    enc.close()


@pytest.mark.parametrize("w,h,q,name", [(240, 2304, 5, "tablelike"), (1920, 1088, 1, "mobilelike"), (2048, 1088, 16, "staticlike")])
def test_tall_frames_dc_chain_bands_as_one_wavefront(w, h, q, name, monkeypatch):
    """Frames taller than 512 lines with 2048 macroblocks or more: the bands of the P frames' DC chains run as waves of one
    continued wavefront (dc_chain_band_wave: five luma and three chroma bands at 2304 lines, three and two at 1088), each waiting
    on the band above through a progress word in LDS.  Against the oracle, and against the band-after-band form
    (ICSP_SERIAL_BANDS=0).  DPCM_DC_block ENC:3822-3988, CDPCM_DC_block ENC:4420-4514."""
    n = 4
    clip = clipgen.synth_clip(name, 1, width=w, height=h).repeat(n, axis=0) if name == "staticlike" else clipgen.synth_clip(name, n, width=w, height=h)
    want = po.encode_sequence(clip, w, h, q, q, 4, nthreads=NT)
    for bands in ("1", "0"):
        monkeypatch.setenv("ICSP_SERIAL_BANDS", bands)
        enc = capi.Encoder(w, h, q, q, 4, max_frames=n)
        enc.upload(clip)
        for _ in range(3):
            enc.encode_resident(0, n)
        got = enc.download(0, n)
        enc.close()
        _cmp(got, want, f"{w}x{h} bands={bands}: ")
