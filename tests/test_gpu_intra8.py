"""The throughput form of the intra luma kernel (k_intra_luma8: eight lanes per block, eight blocks per wave) against the
oracle, forced through ICSP_INTRA_FORM=8 on batches where the default would pick the 32-lane latency form, and the default
choice on a batch large enough to pick it by itself.  Same reference code as the 32-lane form: DPCM_pix_block ENC:851-1499,
DCT_block ENC:2685, DPCM_DC_block ENC:3643, Quantization_block ENC:2750, IDCT_block ENC:2825, IDPCM_pix_block ENC:1500-1875
(ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp)."""
import os

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
KEYS = ("levels", "acflag", "mpm", "mvd", "recon")
NT = min(os.cpu_count() or 1, 64)


def _cmp(got, want, ctx=""):
    for k in KEYS:
        if not np.array_equal(got[k], want[k]):
            bad = np.argwhere(got[k] != want[k])
            raise AssertionError(f"{ctx}{k}: {len(bad)} mismatches, first at {bad[0].tolist()}: "
                                 f"got {got[k][tuple(bad[0])]} want {want[k][tuple(bad[0])]}")


@pytest.fixture
def form8():
    os.environ["ICSP_INTRA_FORM"] = "8"
    yield
    del os.environ["ICSP_INTRA_FORM"]


@pytest.mark.parametrize("name,n,qdc,qac,period,w,h", [
    ("foremanlike", 4, 16, 16, 0, 352, 288), ("mobilelike", 3, 1, 1, 0, 352, 288), ("stefanlike", 6, 8, 8, 3, 352, 288),
    ("mobilelike", 3, 16, 1, 3, 352, 288), ("staticlike", 4, 1, 1, 4, 352, 288), ("tablelike", 3, 8, 8, 0, 64, 48),
    ("newslike", 2, 16, 16, 0, 32, 16), ("stefanlike", 3, 8, 8, 3, 416, 240), ("mobilelike", 2, 8, 8, 0, 704, 576),
    ("tablelike", 3, 16, 16, 3, 1920, 1088), ("mobilelike", 2, 16, 16, 2, 2048, 1088),
    ("mobilelike", 2, 16, 16, 2, 32, 2304), ("tablelike", 2, 8, 8, 0, 4096, 32), ("stefanlike", 2, 1, 1, 0, 48, 1600),
])
def test_forced_8_lane_form_matches_oracle(form8, name, n, qdc, qac, period, w, h):
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, qdc, qac, period, max_frames=n)
    enc.keep_coef(True)
    got = enc.encode(clip)
    coef = enc.download_coef(0, 1)
    mv, mode = enc.download_debug(0, n)
    enc.close()
    want = po.encode_sequence(clip, w, h, qdc, qac, period, nthreads=NT)
    _cmp(got, want, f"{name} {w}x{h} n={n} q={qdc}/{qac} p={period}: ")
    dbg = po.intra_frame(clip[0], w, h, qdc, qac, want_dbg=True)
    assert np.array_equal(coef[0].view(np.int64), dbg["coef"].view(np.int64))        # forward DCT coefficients: 0 ulp
    assert np.array_equal(mode[0], dbg["mode"])


@pytest.mark.parametrize("name,n,qdc,qac,period,w,h", [
    ("foremanlike", 3, 16, 16, 0, 352, 288), ("mobilelike", 2, 1, 1, 0, 352, 288), ("stefanlike", 4, 8, 8, 2, 352, 288),
    ("tablelike", 3, 8, 8, 0, 64, 48), ("newslike", 2, 16, 16, 0, 32, 16), ("mobilelike", 2, 3, 255, 0, 720, 480),
    ("mobilelike", 2, 16, 16, 2, 32, 2304), ("tablelike", 2, 8, 8, 0, 4096, 32), ("mobilelike", 2, 2, 5, 0, 176, 144),
])
def test_rows_chained_in_pairs_matches_oracle(monkeypatch, name, n, qdc, qac, period, w, h):
    """ICSP_INTRA_GROUP=2: block rows chained in pairs (96 steps per CIF frame, four waves)."""
    monkeypatch.setenv("ICSP_INTRA_GROUP", "2")
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, qdc, qac, period, max_frames=n)
    got = enc.encode(clip)
    ch = enc.last_choice()
    enc.close()
    assert (ch["intra_lanes_per_block"], ch["intra_rows_chained"], ch["intra_recon_ring"]) == (8, 2, True)
    _cmp(got, po.encode_sequence(clip, w, h, qdc, qac, period, nthreads=NT), f"{name} {w}x{h} n={n} q={qdc}/{qac} p={period}: ")


@pytest.mark.parametrize("w,h,chained", [(704, 576, 2), (1280, 720, 0), (1920, 1088, 0)])
def test_pairs_fall_back_to_the_plain_wavefront_where_a_step_does_not_fit_eight_waves(form8, w, h, chained):
    clip = clipgen.synth_clip("mobilelike", 2, width=w, height=h)
    enc = capi.Encoder(w, h, 8, 8, 0, max_frames=2)
    got = enc.encode(clip)
    ch = enc.last_choice()
    enc.close()
    assert ch["intra_lanes_per_block"] == 8 and ch["intra_rows_chained"] == chained
    _cmp(got, po.encode_sequence(clip, w, h, 8, 8, 0, nthreads=NT), f"{w}x{h}: ")


def test_rows_chained_reference_clip(monkeypatch):
    """300 frames of the clip the reference CLI was run on, rows in pairs: its reconstruction and .bin hashes."""
    import hashlib
    import json
    monkeypatch.setenv("ICSP_INTRA_GROUP", "2")
    monkeypatch.setenv("ICSP_INTRA_FORM", "8")
    clip = clipgen.synth_clip("foremanlike", 300)
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=300)
    enc.upload(clip)
    for _ in range(3):
        enc.encode_resident(0, 300)
    rec = enc.download(0, 300, what=("recon",))["recon"]
    bs = enc.pack_bitstream(0, 300)
    ch = enc.last_choice()
    enc.close()
    assert ch["intra_rows_chained"] == 2
    ref = next(s for s in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))
               if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
    assert hashlib.sha256(rec.tobytes()).hexdigest() == ref["recon_sha256"]
    assert hashlib.sha256(bs).hexdigest() == ref["bin_sha256"]


@pytest.mark.parametrize("group", ["1", "2"])
@pytest.mark.parametrize("w,h,n,period", [(352, 288, 3, 0), (48, 64, 4, 2), (80, 32, 3, 0), (112, 48, 3, 3), (144, 176, 2, 0), (704, 576, 2, 0)])
def test_8_lane_form_variants(form8, monkeypatch, group, w, h, n, period):
    """The two wavefronts of the 8-lane kernel -- plain (ICSP_INTRA_GROUP=1: the six-wave build, waves idle on small frames) and rows in
    pairs -- on widths whose last piece of a reconstruction row is partial (6, 10, 14, 18 blocks) and full."""
    monkeypatch.setenv("ICSP_INTRA_GROUP", group)
    clip = clipgen.synth_clip("mobilelike", n, width=w, height=h)
    enc = capi.Encoder(w, h, 8, 8, period, max_frames=n)
    got = enc.encode(clip)
    ch = enc.last_choice()
    enc.close()
    assert (ch["intra_lanes_per_block"], ch["intra_recon_ring"]) == (8, True)
    assert ch["intra_rows_chained"] == (2 if group == "2" else 0)
    _cmp(got, po.encode_sequence(clip, w, h, 8, 8, period, nthreads=NT), f"{w}x{h} group={group}: ")


def test_default_picks_the_8_lane_form_on_a_loaded_chip():
    """More than two frames per CU: 600 all-intra CIF frames (the reference clip twice) -> the 300-frame reference hash twice."""
    import hashlib
    import json
    clip = clipgen.synth_clip("foremanlike", 300)
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=600)
    enc.upload(np.concatenate([clip, clip]))
    enc.encode_resident(0, 600)
    rec = enc.download(0, 600, what=("recon",))["recon"]
    bs = enc.pack_bitstream(0, 300)
    enc.close()
    ref = next(s for s in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))
               if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
    assert hashlib.sha256(rec[:300].tobytes()).hexdigest() == ref["recon_sha256"]
    assert hashlib.sha256(rec[300:].tobytes()).hexdigest() == ref["recon_sha256"]
    assert hashlib.sha256(bs).hexdigest() == ref["bin_sha256"]


@pytest.mark.parametrize("parts", ["1", "2"])
def test_all_intra_batch_in_two_parts_back_to_back(parts):
    """More frames than CUs: the luma kernel goes out as two launches on two streams (ICSP_I_GROUPS, encode_range), and
    passes over the same range follow each other part by part without a join.  Back-to-back passes, a smaller range on the
    same slots in between (one launch again), new input, and a context that shares the device."""
    import hashlib
    import json
    os.environ["ICSP_I_GROUPS"] = parts
    try:
        clip = clipgen.synth_clip("foremanlike", 300)
        ref = next(s for s in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))
                   if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
        enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=300)
        other_ctx = capi.Encoder(352, 288, 16, 16, 0, max_frames=300)
        enc.upload(clip)
        other_ctx.upload(clip)
        for _ in range(5):
            enc.encode_resident(0, 300)
            other_ctx.encode_resident(0, 300)

        def check(e, what):
            rec = e.download(0, 300, what=("recon",))["recon"]
            assert hashlib.sha256(rec.tobytes()).hexdigest() == ref["recon_sha256"], what
            assert hashlib.sha256(e.pack_bitstream(0, 300)).hexdigest() == ref["bin_sha256"], what
        check(enc, "five passes")
        check(other_ctx, "five passes, second context")
        other_ctx.close()
        enc.encode_resident(0, 100)
        enc.encode_resident(0, 300)
        enc.encode_resident(0, 300)
        check(enc, "after a smaller range")
        other = clipgen.synth_clip("mobilelike", 300)
        enc.upload(other)
        enc.encode_resident(0, 300)
        enc.encode_resident(0, 300)
        got = enc.download(0, 300)
        enc.close()
    finally:
        del os.environ["ICSP_I_GROUPS"]
    _cmp(got, po.encode_sequence(other, 352, 288, 16, 16, 0, nthreads=NT), f"parts={parts}, new input: ")


@pytest.mark.parametrize("var,val", [("ICSP_I_GROUPS", "3"), ("ICSP_P_GROUPS", "0"), ("ICSP_NO_FUSE", "yes"), ("ICSP_INTRA_FORM", "16"),
                                     ("ICSP_INTRA_GROUP", "3"), ("ICSP_INTRA_GROUP", "4"), ("ICSP_XCD_SLICES", "-1"), ("ICSP_I_GROUPS", ""), ("ICSP_QUANT_POW2", "2"), ("ICSP_CHROMA_CAP", "121")])
def test_override_outside_its_range_fails_the_create(var, val):
    """include/icsp_hip.h: a tuning override that is not a whole number in its range makes icsp_create fail."""
    os.environ[var] = val
    try:
        with pytest.raises(RuntimeError):
            capi.Encoder(352, 288, 16, 16, 0, max_frames=2)
    finally:
        del os.environ[var]

