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


@pytest.fixture
def rows4(monkeypatch):
    monkeypatch.setenv("ICSP_INTRA_GROUP", "4")


@pytest.mark.parametrize("name,n,qdc,qac,period,w,h", [
    ("foremanlike", 4, 16, 16, 0, 352, 288), ("mobilelike", 3, 1, 1, 0, 352, 288), ("stefanlike", 6, 8, 8, 3, 352, 288),
    ("mobilelike", 3, 16, 1, 3, 352, 288), ("staticlike", 4, 1, 1, 4, 352, 288), ("tablelike", 3, 8, 8, 0, 64, 48),
    ("newslike", 2, 16, 16, 0, 32, 16), ("stefanlike", 3, 8, 8, 3, 416, 240), ("mobilelike", 2, 3, 255, 0, 720, 480),
    ("mobilelike", 2, 16, 16, 2, 32, 2304), ("tablelike", 2, 8, 8, 0, 4096, 32), ("stefanlike", 2, 1, 1, 0, 48, 1600),
    ("mobilelike", 2, 2, 5, 0, 176, 144), ("foremanlike", 2, 31, 2, 0, 352, 80), ("mobilelike", 2, 64, 64, 0, 1024, 512),
])
def test_rows_chained_in_fours_matches_oracle(rows4, name, n, qdc, qac, period, w, h):
    """k_intra_luma8<.., 4>: block rows one wavefront step apart in groups of four, the DC predictors that need the upper-right
    neighbour of the same step resolved through the lanes (DPCM_DC_block's median, ENC:3652-3818).  Every output and the forward
    coefficients (0 ulp) against the oracle; the geometries include one-wave frames, eight-wave frames and tall narrow ones."""
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, qdc, qac, period, max_frames=n)
    enc.keep_coef(True)
    got = enc.encode(clip)
    ch = enc.last_choice()
    coef = enc.download_coef(0, 1)
    mv, mode = enc.download_debug(0, n)
    enc.close()
    assert (ch["intra_lanes_per_block"], ch["intra_rows_chained"], ch["intra_recon_ring"]) == (8, 4, True)
    want = po.encode_sequence(clip, w, h, qdc, qac, period, nthreads=NT)
    _cmp(got, want, f"{name} {w}x{h} n={n} q={qdc}/{qac} p={period}: ")
    dbg = po.intra_frame(clip[0], w, h, qdc, qac, want_dbg=True)
    assert np.array_equal(coef[0].view(np.int64), dbg["coef"].view(np.int64))
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


@pytest.mark.parametrize("w,h", [(704, 576), (1920, 1088)])
def test_rows_chained_falls_back_where_a_step_does_not_fit_eight_waves(rows4, w, h):
    clip = clipgen.synth_clip("mobilelike", 2, width=w, height=h)
    enc = capi.Encoder(w, h, 8, 8, 0, max_frames=2)
    got = enc.encode(clip)
    ch = enc.last_choice()
    enc.close()
    assert ch["intra_rows_chained"] == 0
    _cmp(got, po.encode_sequence(clip, w, h, 8, 8, 0, nthreads=NT), f"{w}x{h}: ")


@pytest.mark.parametrize("nw", ["6", "8"])
def test_rows_chained_with_idle_waves(rows4, nw, monkeypatch):
    monkeypatch.setenv("ICSP_INTRA_NW", nw)
    clip = clipgen.synth_clip("foremanlike", 2)
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=2)
    got = enc.encode(clip)
    ch = enc.last_choice()
    enc.close()
    assert ch["intra_rows_chained"] == 4 and ch["intra_waves_per_workgroup"] == int(nw)
    _cmp(got, po.encode_sequence(clip, 352, 288, 16, 16, 0), f"NW={nw}: ")


def test_rows_chained_reference_clip(rows4):
    """300 frames of the clip the reference CLI was run on: its reconstruction and .bin hashes."""
    import hashlib
    import json
    clip = clipgen.synth_clip("foremanlike", 300)
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=300)
    enc.upload(clip)
    for _ in range(3):
        enc.encode_resident(0, 300)
    rec = enc.download(0, 300, what=("recon",))["recon"]
    bs = enc.pack_bitstream(0, 300)
    ch = enc.last_choice()
    enc.close()
    assert ch["intra_rows_chained"] == 4
    ref = next(s for s in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "streams.json")))
               if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
    assert hashlib.sha256(rec.tobytes()).hexdigest() == ref["recon_sha256"]
    assert hashlib.sha256(bs).hexdigest() == ref["bin_sha256"]


@pytest.mark.parametrize("ring", ["0", "1"])
@pytest.mark.parametrize("w,h,n,period", [(352, 288, 3, 0), (48, 64, 4, 2), (80, 32, 3, 0), (112, 48, 3, 3), (144, 176, 2, 0), (704, 576, 2, 0)])
def test_8_lane_form_variants(form8, ring, w, h, n, period):
    """The two builds of the kernel -- reconstruction through the LDS ring or straight out (ICSP_INTRA_RING) -- on widths whose
    last piece of a row is partial (6, 10, 14, 18 blocks) and full."""
    os.environ["ICSP_INTRA_RING"] = ring
    try:
        clip = clipgen.synth_clip("mobilelike", n, width=w, height=h)
        enc = capi.Encoder(w, h, 8, 8, period, max_frames=n)
        got = enc.encode(clip)
        ch = enc.last_choice()
        enc.close()
    finally:
        del os.environ["ICSP_INTRA_RING"]
    assert (ch["intra_lanes_per_block"], ch["intra_recon_ring"]) == (8, ring == "1")
    _cmp(got, po.encode_sequence(clip, w, h, 8, 8, period, nthreads=NT), f"{w}x{h} ring={ring}: ")


@pytest.mark.parametrize("nw", ["1", "2", "16"])
def test_8_lane_form_any_workgroup_width(form8, nw):
    """Fewer waves than the widest step needs (several rounds per step) and more than it needs (idle waves)."""
    os.environ["ICSP_INTRA_NW"] = nw
    try:
        clip = clipgen.synth_clip("foremanlike", 2)
        enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=2)
        got = enc.encode(clip)
        enc.close()
    finally:
        del os.environ["ICSP_INTRA_NW"]
    _cmp(got, po.encode_sequence(clip, 352, 288, 16, 16, 0), f"NW={nw}: ")


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
                                     ("ICSP_INTRA_NW", "17"), ("ICSP_INTRA_GROUP", "3"), ("ICSP_XCD_SLICES", "-1"), ("ICSP_I_GROUPS", ""), ("ICSP_INTRA_RING", "2"), ("ICSP_CHROMA_CAP", "121")])
def test_override_outside_its_range_fails_the_create(var, val):
    """include/icsp_hip.h: a tuning override that is not a whole number in its range makes icsp_create fail."""
    os.environ[var] = val
    try:
        with pytest.raises(RuntimeError):
            capi.Encoder(352, 288, 16, 16, 0, max_frames=2)
    finally:
        del os.environ[var]


# ---- half-frame units (ICSP_INTRA_SPLIT=1, k_intra_luma8s): a frame as two workgroups, the lower half taking the row above its first
#      row from the upper half through tagged granules in device memory
@pytest.mark.parametrize("name,n,qdc,qac,period,w,h", [
    ("foremanlike", 5, 16, 16, 0, 352, 288), ("mobilelike", 3, 1, 1, 0, 352, 288), ("stefanlike", 6, 8, 8, 3, 352, 288),
    ("tablelike", 3, 8, 8, 0, 64, 48), ("newslike", 3, 16, 16, 0, 32, 32), ("mobilelike", 2, 3, 255, 0, 720, 480),
    ("mobilelike", 2, 16, 16, 2, 32, 2304), ("tablelike", 2, 8, 8, 0, 4096, 32), ("mobilelike", 3, 2, 5, 0, 176, 144),
    ("foremanlike", 3, 31, 2, 0, 352, 80), ("stefanlike", 2, 16, 16, 0, 352, 576), ("mobilelike", 2, 64, 64, 0, 1024, 512),
])
def test_half_frame_units_match_oracle(monkeypatch, name, n, qdc, qac, period, w, h):
    monkeypatch.setenv("ICSP_INTRA_GROUP", "2")
    monkeypatch.setenv("ICSP_INTRA_SPLIT", "1")
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, qdc, qac, period, max_frames=n)
    got = enc.encode(clip)
    ch, split = enc.last_choice(), enc.last_split()
    mv, mode = enc.download_debug(0, n)
    enc.close()
    assert (ch["intra_lanes_per_block"], ch["intra_rows_chained"]) == (8, 2)
    assert split == 1
    _cmp(got, po.encode_sequence(clip, w, h, qdc, qac, period, nthreads=NT), f"{name} {w}x{h} n={n} q={qdc}/{qac} p={period}: ")
    assert np.array_equal(mode[0], po.intra_frame(clip[0], w, h, qdc, qac, want_dbg=True)["mode"])


def test_half_frame_units_two_rows_only_falls_back(monkeypatch):
    """32x16: two block rows -- no pair for each half; the launch is the one-workgroup form."""
    monkeypatch.setenv("ICSP_INTRA_GROUP", "2")
    monkeypatch.setenv("ICSP_INTRA_SPLIT", "1")
    clip = clipgen.synth_clip("newslike", 2, width=32, height=16)
    enc = capi.Encoder(32, 16, 16, 16, 0, max_frames=2)
    got = enc.encode(clip)
    split = enc.last_split()
    enc.close()
    assert split == 0
    _cmp(got, po.encode_sequence(clip, 32, 16, 16, 16, 0, nthreads=NT), "32x16: ")


def test_half_frame_units_under_load_match_the_one_workgroup_form(monkeypatch):
    """The hand-off under the conditions that expose a wrong protocol (uneven load, consumers whose caches have seen the lines of
    earlier passes): two ranges of 300 + 340 frames encoded in turn, forty passes without a host wait in between, more units in a
    launch than the chip holds at once on the larger batch; every pass's results are the one-workgroup form's, checked after the
    last pass of each range and after the first."""
    n0, n1 = 300, 1500
    clip = np.concatenate([clipgen.synth_clip("foremanlike", 300), clipgen.synth_clip("mobilelike", 60)] * 5)[: n0 + n1]
    ref = capi.Encoder(352, 288, 16, 16, 0, max_frames=n0 + n1)
    ref.upload(clip)
    ref.encode_resident(0, n0 + n1)
    want = ref.download(0, n0 + n1, what=("levels", "recon", "mpm"))
    ref.close()
    monkeypatch.setenv("ICSP_INTRA_FORM", "8")
    monkeypatch.setenv("ICSP_INTRA_GROUP", "2")
    monkeypatch.setenv("ICSP_INTRA_SPLIT", "1")
    enc = capi.Encoder(352, 288, 16, 16, 0, max_frames=n0 + n1)
    enc.upload(clip)
    for rep in range(40):
        enc.encode_resident(0, n0)
        enc.encode_resident(n0, n1)
        if rep in (0, 39):
            assert enc.last_split() == 1
            got = enc.download(0, n0 + n1, what=("levels", "recon", "mpm"))
            for k in got:
                assert np.array_equal(got[k], want[k]), (rep, k)
    enc.close()
