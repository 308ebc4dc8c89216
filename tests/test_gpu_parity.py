"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed reference fixtures.
Integer outputs are bit-exact; DCT coefficients are compared bit-for-bit too (tolerance allowed by the
north star is 1e-4, asserted separately)."""
import glob
import hashlib
import json
import os
import re

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
W, H = 352, 288
KEYS = ("levels", "acflag", "mpm", "mvd", "recon")


def _cmp(got, want, ctx=""):
    for k in KEYS:
        if not np.array_equal(got[k], want[k]):
            bad = np.argwhere(got[k] != want[k])
            raise AssertionError(f"{ctx}{k}: {len(bad)} mismatches, first at {bad[0].tolist()}: "
                                 f"got {got[k][tuple(bad[0])]} want {want[k][tuple(bad[0])]}")


@pytest.mark.parametrize("name,n,q,period", [
    ("foremanlike", 2, 16, 0), ("foremanlike", 3, 8, 0), ("mobilelike", 2, 1, 0),
    ("stefanlike", 3, 8, 3), ("foremanlike", 6, 16, 3), ("footballlike", 4, 16, 4),
    ("staticlike", 4, 1, 4), ("akiyolike", 5, 16, 5), ("tablelike", 7, 8, 3),
    ("staticlike", 36, 16, 6),          # six GOPs whose every P frame has early breaks: the fused kernel's waiting path
])
def test_sequence_matches_oracle(name, n, q, period):
    clip = clipgen.synth_clip(name, n)
    enc = capi.Encoder(W, H, q, q, period, max_frames=n)
    got = enc.encode(clip)
    enc.close()
    want = po.encode_sequence(clip, W, H, q, q, period)
    _cmp(got, want, f"{name} n={n} q={q} p={period}: ")


def test_split_qp():
    clip = clipgen.synth_clip("mobilelike", 3)
    enc = capi.Encoder(W, H, 16, 1, 3, max_frames=3)
    got = enc.encode(clip)
    enc.close()
    _cmp(got, po.encode_sequence(clip, W, H, 16, 1, 3))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "frames_*.npz"))),
                         ids=lambda p: os.path.basename(p)[7:-4])
def test_reference_fixtures(path):
    """Straight against what the compiled reference produced (levels, flags, mvd, intra modes, raw mv, recon)."""
    m = re.match(r"frames_(\w+?)_(\d+)f_q(\d+)_(\d+)_p(\d+)\.npz", os.path.basename(path))
    name, n, qdc, qac, period = m.group(1), *map(int, m.groups()[1:])
    g = np.load(path)
    clip = clipgen.synth_clip(name, n)
    enc = capi.Encoder(W, H, qdc, qac, period, max_frames=n)
    got = enc.encode(clip)
    mv, mode = enc.download_debug(0, n)
    enc.close()
    for k in ("levels", "acflag", "mpm", "mvd"):
        assert np.array_equal(got[k], g[k]), k
    assert hashlib.sha256(got["recon"].tobytes()).hexdigest() == str(g["recon_sha"])
    assert np.array_equal(mode, g["mode"]) and np.array_equal(mv, g["mv"])


def test_dct_coefficients_bit_exact_and_within_tolerance():
    clip = clipgen.synth_clip("stefanlike", 3)
    enc = capi.Encoder(W, H, 8, 8, 3, max_frames=3)
    enc.keep_coef(True)
    enc.encode(clip)
    coef = enc.download_coef(0, 3)
    enc.close()
    ref = po.encode_sequence(clip, W, H, 8, 8, 3)
    r0 = po.intra_frame(clip[0], W, H, 8, 8, want_dbg=True)["coef"]
    r1 = po.inter_frame(clip[1], ref["recon"][0], W, H, 8, 8, want_dbg=True)["coef"]
    for got, want in ((coef[0], r0), (coef[1], r1)):
        assert np.max(np.abs(got - want)) <= 1e-4                          # the stated tolerance
        assert np.array_equal(got.view(np.int64), want.view(np.int64))     # and in fact 0 ulp


def test_early_break_state_carry_pair(golden_dir):
    """Crafted frame pair with exact-match blocks: SAD==0 early breaks change the search state of later MBs."""
    g = np.load(os.path.join(golden_dir, "me_static_pair.npz"))
    cur, prev = g["cur"], g["prev"]
    clip = np.zeros((2, W * H * 3 // 2), np.uint8)
    clip[:, W * H:] = 128
    clip[0, :W * H] = prev.ravel()
    clip[1, :W * H] = cur.ravel()
    enc = capi.Encoder(W, H, 1, 1, 2, max_frames=2)
    got = enc.encode(clip)
    mv, _ = enc.download_debug(0, 2)
    enc.close()
    want = po.encode_sequence(clip, W, H, 1, 1, 2)
    _cmp(got, want)
    mx, my, ns = po.me_frame(cur, want["recon"][0, :W * H].reshape(H, W))
    assert np.array_equal(mv[1, :, 0], mx) and np.array_equal(mv[1, :, 1], my)
    assert (ns < 64).any()


def test_resident_api_and_gop_concurrency():
    """upload / encode_resident / download on 4 GOPs at once equals GOP-by-GOP oracle output."""
    clip = clipgen.synth_clip("coastguardlike", 12)
    enc = capi.Encoder(W, H, 16, 16, 3, max_frames=12)
    enc.upload(clip, 0)
    enc.encode_resident(0, 12)
    got = enc.download(0, 12)
    enc.encode_resident(6, 6)          # a GOP-aligned sub-range re-encoded in place gives the same answer
    again = enc.download(6, 6)
    enc.close()
    want = po.encode_sequence(clip, W, H, 16, 16, 3, nthreads=4)
    _cmp(got, want)
    for k in KEYS:
        assert np.array_equal(again[k], want[k][6:])


def test_ragged_last_gop_and_single_frame():
    clip = clipgen.synth_clip("newslike", 5)
    enc = capi.Encoder(W, H, 16, 16, 3, max_frames=5)
    got = enc.encode(clip)            # GOPs of 3 + 2
    one = enc.encode(clip[:1])
    enc.close()
    want = po.encode_sequence(clip, W, H, 16, 16, 3)
    _cmp(got, want)
    for k in KEYS:
        assert np.array_equal(one[k][0], want[k][0])


def test_other_geometry_non_cif():
    """The reference hard-codes 352x288 (encoder_main.cpp:20); the kernels take any multiple of 16."""
    for (w, h) in ((64, 48), (416, 240)):
        clip = clipgen.synth_clip("stefanlike", 3, width=w, height=h)
        enc = capi.Encoder(w, h, 8, 8, 3, max_frames=3)
        got = enc.encode(clip)
        enc.close()
        _cmp(got, po.encode_sequence(clip, w, h, 8, 8, 3), f"{w}x{h}: ")


def test_hd_1088p_config5_geometry():
    """BASELINE configs[4] geometry: 1920x1088 (1080 is not a multiple of 16, ENC:322), --intraPeriod 3 here.  Exercises the
    paths CIF never takes: more wavefront blocks than one workgroup round (60 waves needed, 16 used), block rows > 64 in the
    DC chains (LDS fallback), 8160 macroblocks per frame in the per-frame serial kernel."""
    w, h = 1920, 1088
    clip = clipgen.synth_clip("tablelike", 3, width=w, height=h)
    enc = capi.Encoder(w, h, 16, 16, 3, max_frames=3)
    got = enc.encode(clip)
    enc.close()
    _cmp(got, po.encode_sequence(clip, w, h, 16, 16, 3, nthreads=3), "1088p: ")


@pytest.mark.parametrize("env", [{}, {"ICSP_NO_FUSE": "1"}, {"ICSP_P_GROUPS": "1"}], ids=["fused", "two-launch", "one-group"])
def test_many_small_gops_one_call(env):
    """1100 GOPs in one call (tiny frames, 2200 of them, GOPs of 2; a repeated frame inside every fourth GOP raises the
    early-break flags): the fused P step (k_serial_fused: per-frame serial workgroups + four-state search, last arriver runs
    the serial body), the same as two launches (ICSP_NO_FUSE=1: k_me<true,16> + k_frame_serial), and one GOP group."""
    w, h, n = 64, 48, 2200
    base = clipgen.synth_clip("stefanlike", 40, width=w, height=h)
    clip = np.concatenate([base] * (n // 40))
    clip[1::8] = clip[0::8]                          # every fourth GOP: P frame identical to its I frame
    os.environ.update(env)
    try:
        enc = capi.Encoder(w, h, 16, 16, 2, max_frames=n)
    finally:
        for k in env:
            del os.environ[k]
    got = enc.encode(clip)
    bs = enc.pack_bitstream(0, n)
    enc.close()
    want = po.encode_sequence(clip, w, h, 16, 16, 2, nthreads=8)
    _cmp(got, want, "2200 x 64x48: ")
    assert bs == capi.write_bitstream(w, h, 16, 16, 2, want["levels"], want["acflag"], want["mpm"], want["mvd"])


def test_largest_supported_frame_2048x1088():
    """Exactly the 8704-macroblock limit of the header: the per-frame kernels stage a whole frame's chain inputs in LDS
    (k_frame_serial 130 KB, k_dec_serial 139 KB + 16 KB static of the 160 KB).  I + P, encode, device packer, decode."""
    w, h = 2048, 1088
    clip = clipgen.synth_clip("mobilelike", 2, width=w, height=h)
    enc = capi.Encoder(w, h, 16, 16, 2, max_frames=2)
    got = enc.encode(clip)
    bs = enc.pack_bitstream(0, 2)
    enc.decode_resident(0, 2)
    dec = enc.download(0, 2, what=("recon",))["recon"]
    enc.close()
    want = po.encode_sequence(clip, w, h, 16, 16, 2, nthreads=2)
    _cmp(got, want, "2048x1088: ")
    assert bs == capi.write_bitstream(w, h, 16, 16, 2, want["levels"], want["acflag"], want["mpm"], want["mvd"])
    assert np.array_equal(dec, po.decode_sequence(want["levels"], want["mpm"], want["mvd"], w, h, 16, 16, 2))


@pytest.mark.parametrize("w,h,period,q", [(4096, 16, 2, 16), (4096, 32, 0, 8), (4096, 32, 3, 1), (4096, 528, 2, 16),
                                          (32, 2304, 2, 16), (32, 2304, 0, 1), (48, 1600, 3, 8), (64, 2176, 2, 16)])
def test_extreme_aspect_ratios(w, h, period, q):
    """The widest frame the interface takes (4096: 512 blocks on a row, one to 33 macroblock rows) and the tallest (2304 lines:
    288 block rows, the DC chains in five bands of 64 rows; motion search with two macroblock columns).  Encode (forced 8-lane
    form as well where the frame is all-intra), device packer, decode."""
    n = 3
    clip = clipgen.synth_clip("mobilelike", n, width=w, height=h)
    want = po.encode_sequence(clip, w, h, q, q, period, nthreads=4)
    enc = capi.Encoder(w, h, q, q, period, max_frames=n)
    got = enc.encode(clip)
    bs = enc.pack_bitstream(0, n)
    enc.decode_resident(0, n)
    dec = enc.download(0, n, what=("recon",))["recon"]
    enc.close()
    _cmp(got, want, f"{w}x{h} p={period} q={q}: ")
    assert bs == capi.write_bitstream(w, h, q, q, period, want["levels"], want["acflag"], want["mpm"], want["mvd"])
    assert np.array_equal(dec, po.decode_sequence(want["levels"], want["mpm"], want["mvd"], w, h, q, q, period))


def test_4cif_all_intra_and_size_limit():
    w, h = 704, 576
    clip = clipgen.synth_clip("mobilelike", 2, width=w, height=h)
    enc = capi.Encoder(w, h, 8, 8, 0, max_frames=2)
    got = enc.encode(clip)
    enc.close()
    _cmp(got, po.encode_sequence(clip, w, h, 8, 8, 0), "4CIF: ")
    with pytest.raises(capi.IcspError):
        capi.Encoder(4096, 2304)                    # 36864 macroblocks: beyond the 8704-macroblock limit of the header


def test_errors_not_exit():
    with pytest.raises(capi.IcspError):
        capi.Encoder(350, 288)                      # width not a multiple of 16 (ENC:322-326 returns -1)
    with pytest.raises(capi.IcspError):
        capi.Encoder(352, 288, 0, 16)               # QP 0 would divide by zero in the reference
    enc = capi.Encoder(W, H, 16, 16, 3, max_frames=3)
    with pytest.raises(capi.IcspError):
        enc.encode_resident(1, 2)                   # not GOP aligned
    with pytest.raises(capi.IcspError):
        enc.encode_resident(0, 4)                   # beyond capacity
    enc.close()


def test_bitstream_bit_identical_to_reference_small(golden_dir):
    for fn, name, n, q, period in (("foremanlike_2f_q16_p0.bin", "foremanlike", 2, 16, 0),
                                   ("stefanlike_3f_q8_p3.bin", "stefanlike", 3, 8, 3)):
        clip = clipgen.synth_clip(name, n)
        enc = capi.Encoder(W, H, q, q, period, max_frames=n)
        o = enc.encode(clip)
        enc.close()
        bs = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
        assert bs == open(os.path.join(golden_dir, fn), "rb").read()


def test_full_baseline_configs_hashes(golden_dir):
    """BASELINE configs 2 and 3 at full length: .bin and test_yuv.yuv SHA-256 equal the reference's."""
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    want = {("foremanlike", 300, 16, 0), ("stefanlike", 300, 8, 10)}
    seen = 0
    for s in streams:
        key = (s["clip"], s["nframes"], s["qp"], s["intra_period"])
        if key not in want or "bin_sha256" not in s:
            continue
        clip = clipgen.synth_clip(s["clip"], s["nframes"])
        enc = capi.Encoder(W, H, s["qp"], s["qp"], s["intra_period"], max_frames=s["nframes"])
        o = enc.encode(clip)
        enc.close()
        assert hashlib.sha256(o["recon"].tobytes()).hexdigest() == s["recon_sha256"], key
        bs = capi.write_bitstream(W, H, s["qp"], s["qp"], s["intra_period"], o["levels"], o["acflag"], o["mpm"], o["mvd"])
        assert len(bs) == s["bin_bytes"] and hashlib.sha256(bs).hexdigest() == s["bin_sha256"], key
        seen += 1
    assert seen == 2


def test_config4_all_twelve_clips_hashes(golden_dir):
    """BASELINE configs[3]: the twelve CIF clips (11 x 300 f + 1 x 90 f), --intraPeriod 10, QP 16.  Each clip's .bin and
    test_yuv.yuv SHA-256 equal what the reference CLI produced (tools/make_golden.py)."""
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    todo = [s for s in streams if s["intra_period"] == 10 and s["qp"] == 16 and s["nframes"] >= 90 and "bin_sha256" in s]
    assert len(todo) == 12
    enc = capi.Encoder(W, H, 16, 16, 10, max_frames=300)
    for s in todo:
        clip = clipgen.synth_clip(s["clip"], s["nframes"])
        assert hashlib.sha256(clip.tobytes()).hexdigest() == s["clip_sha256"]
        o = enc.encode(clip)
        assert hashlib.sha256(o["recon"].tobytes()).hexdigest() == s["recon_sha256"], s["clip"]
        bs = capi.write_bitstream(W, H, 16, 16, 10, o["levels"], o["acflag"], o["mpm"], o["mvd"])
        assert len(bs) == s["bin_bytes"] and hashlib.sha256(bs).hexdigest() == s["bin_sha256"], s["clip"]
    enc.close()


@pytest.mark.parametrize("name,n,q,period,w,h", [
    ("foremanlike", 4, 16, 0, W, H), ("stefanlike", 7, 8, 3, W, H), ("mobilelike", 3, 1, 3, W, H),
    ("staticlike", 4, 1, 4, W, H), ("akiyolike", 5, 16, 5, W, H), ("tablelike", 3, 8, 2, 64, 48), ("newslike", 2, 16, 2, 32, 16),
])
def test_device_bit_packer_equals_host_writer(name, n, q, period, w, h):
    """icsp_pack_bits + icsp_bitstream_assemble == icsp_write_bitstream (itself byte-identical to the reference) — large
    codes (QP 1), ACflag runs (static), P headers, a frame of two macroblocks (units of several frames in one wave)."""
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    enc = capi.Encoder(w, h, q, q, period, max_frames=n)
    o = enc.encode(clip)
    want = capi.write_bitstream(w, h, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    got = enc.pack_bitstream(0, n)
    assert len(got) == len(want)
    assert got == want
    L = period if period else 1
    if n > L:                                           # a GOP-aligned sub-range, and shard pieces concatenated on the host
        tail = enc.pack_bitstream(L, n - L)
        s = slice(L, n)
        assert tail == capi.write_bitstream(w, h, q, q, period, o["levels"][s], o["acflag"][s], o["mpm"][s], o["mvd"][s])
        pieces = [enc.pack_bits(0, L), enc.pack_bits(L, n - L)]
        assert capi.assemble_bitstream(w, h, q, q, period, pieces) == want
    enc.close()


@pytest.mark.parametrize("name,n,q,period,cuts", [("foremanlike", 12, 16, 6, [6]), ("staticlike", 9, 16, 3, [3, 6]),
                                                   ("foremanlike", 8, 1, 0, [1, 2, 3, 5]), ("stefanlike", 20, 8, 10, [10])])
def test_pack_into_builds_the_image_in_place(name, n, q, period, cuts):
    """icsp_pack_count + icsp_pack_into from two contexts, batch by batch, straight into ONE zero-initialised image at the
    running bit offset (odd byte and bit phases, strings shorter than the 64-byte alignment window included) + header +
    final byte == icsp_write_bitstream; and a repeated icsp_prepare in between does not disturb anything."""
    clip = clipgen.synth_clip(name, n)
    ref = capi.Encoder(W, H, q, q, period, max_frames=n)
    o = ref.encode(clip)
    want = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    ref.close()
    bounds = [0] + cuts + [n]
    cmax = max(b - a for a, b in zip(bounds, bounds[1:]))
    encs = [capi.Encoder(W, H, q, q, period, max_frames=cmax) for _ in range(2)]
    for e in encs:
        e.prepare()
    for offset in (0, 3):                                # image at an odd address as well (the alignment logic follows it)
        buf = np.zeros(len(want) + 64 + offset, np.uint8)
        image = buf[offset:]
        at = 0
        for k, (a, b) in enumerate(zip(bounds, bounds[1:])):
            e = encs[k % 2]
            e.upload(clip[a:b], 0)
            e.encode_resident(0, b - a)
            bits = e.pack_count(0, b - a)
            e.pack_into(0, b - a, at, image[14:])
            at += bits
            if k == 0:
                e.prepare()
        got = capi.finish_image(W, H, q, q, period, image, at)
        assert len(got) == len(want)
        assert got == want
    with pytest.raises(RuntimeError):                    # pack_into without a count of that range
        encs[0].pack_into(0, 1, 0, np.zeros(1 << 20, np.uint8))
    for e in encs:
        e.close()


def test_device_bit_packer_full_baseline_hash(golden_dir):
    """BASELINE configs[2] (stefanlike 300 f, --intraPeriod 10, QP 8): device-packed .bin SHA-256 == the reference CLI's."""
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    s = next(x for x in streams if (x["clip"], x["nframes"], x["qp"], x["intra_period"]) == ("stefanlike", 300, 8, 10))
    clip = clipgen.synth_clip("stefanlike", 300)
    enc = capi.Encoder(W, H, 8, 8, 10, max_frames=300)
    enc.upload(clip, 0)
    enc.encode_resident(0, 300)
    bs = enc.pack_bitstream(0, 300)
    enc.close()
    assert len(bs) == s["bin_bytes"] and hashlib.sha256(bs).hexdigest() == s["bin_sha256"]
