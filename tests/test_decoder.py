"""Decoder restatement (oracle/icsp_oracle_dec.c) pinned against the reference's own decoder:
  * tests/golden/decoded.json — SHA-256 of what the reference decoder binary wrote for streams of the reference encoder
    (tools/make_golden.py --only decoded); usable on the GPU box where /root/reference does not exist;
  * the binary itself (oracle/_ref/icsp_ref_dec) where it has been built.
The stream is regenerated through the encoder oracle + the product's host packer, both byte-identical to the reference
encoder (test_host_cpu.py), and its SHA-256 is checked against the fixture's before it is decoded."""
import hashlib
import json
import os

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen
from oracle import pyoracle as po

W, H = 352, 288


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "decoded.json")))


def _stream(case):
    name, n, q, period = case["clip"], case["nframes"], case["qp"], case["intra_period"]
    clip = clipgen.synth_clip(name, n)
    o = po.encode_sequence(clip, W, H, q, q, period, nthreads=4)
    bs = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    assert hashlib.sha256(bs).hexdigest() == case["bin_sha256"]
    return clip, o, bs


def test_oracle_decoder_matches_reference_decoder_hashes(golden_dir):
    for case in _cases(golden_dir):
        clip, o, bs = _stream(case)
        n = case["nframes"]
        assert po.parse_header(bs) == (W, H, case["qp"], case["qp"], case["intra_period"])
        p = po.parse_bitstream(bs, n)
        # the parser inverts the packer — except inside the stream's final byte, which the reference writes right-aligned
        # (ENC:4956) and reads MSB-first (DEC:64-70): the last block's tail is garbage in the reference pair itself
        for k in ("acflag", "mpm", "mvd"):
            assert np.array_equal(p[k].reshape(-1)[:-6], o[k].reshape(-1)[:-6]), (case["clip"], k)
        assert np.array_equal(p["levels"][:, :-1], o["levels"][:, :-1])
        dec = po.decode_sequence(p["levels"], p["mpm"], p["mvd"], W, H, case["qp"], case["qp"], case["intra_period"])
        assert hashlib.sha256(dec.tobytes()).hexdigest() == case["decoded_sha256"], case
        assert abs(po.psnr_y(clip, dec, W, H) - case["psnr"]) < 1e-4 + 5e-5          # the reference prints 4 decimals
        # encoder reconstruction vs decoder output: within a grey level or two on a few pixels (float vs double table)
        d = np.abs(dec[:-1].astype(int) - o["recon"][:-1].astype(int))
        assert d.max() <= 2 and (d > 0).mean() < 0.02


@pytest.mark.skipif(not os.path.exists(po.REF_DEC), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,n,q,period", [("tablelike", 5, 16, 1), ("footballlike", 6, 8, 3), ("newslike", 4, 1, 2),
                                             ("coastguardlike", 7, 16, 7)])
def test_oracle_decoder_matches_reference_decoder_binary(tmp_path, name, n, q, period):
    clip = clipgen.synth_clip(name, n)
    o = po.encode_sequence(clip, W, H, q, q, period)
    bs = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    ref, psnr = po.run_ref_decoder(bs, clip, n, q, q, period, str(tmp_path))
    dec = po.decode_bitstream(bs, n)
    assert np.array_equal(dec, ref)
    assert abs(po.psnr_y(clip, dec, W, H) - psnr) < 1e-4


def test_decoder_table_is_double_not_float():
    """DEC.h:19-27 vs ENC.h:190-198: same six-digit literals, but double here and float there."""
    enc = po.costable()
    L = po.lib()
    dec = np.zeros(64, np.float64)
    L.icsp_oracle_dec_costable(dec.ctypes.data_as(__import__("ctypes").c_void_p))
    dec = dec.reshape(8, 8)
    assert dec[1, 0] == 0.980785 and enc[1][0] == float(np.float32(0.980785))
    assert np.abs(dec - np.asarray(enc).reshape(8, 8)).max() < 1e-7 and not np.array_equal(dec, np.asarray(enc).reshape(8, 8))
    blk = (np.arange(64).reshape(8, 8) * 7 % 41 - 20).astype(np.int32)
    assert np.abs(po.dec_idct8x8(blk) - po.idct8x8(blk)).max() < 1e-4


def test_host_parser_inverts_the_packer_and_matches_oracle_parser(golden_dir):
    """Product host parser (icsp_parse_bitstream) vs the oracle's, on reference-identical streams; errors are status codes."""
    case = _cases(golden_dir)[1]
    clip, o, bs = _stream(case)
    n = case["nframes"]
    p, got = capi.parse_bitstream(bs, n)
    assert (p.width, p.height, p.qp_dc, p.qp_ac, p.intra_period) == (W, H, case["qp"], case["qp"], case["intra_period"])
    want = po.parse_bitstream(bs, n)
    for k in ("levels", "acflag", "mpm", "mvd"):
        assert np.array_equal(got[k], want[k]), k
    with pytest.raises(capi.IcspError):
        capi.parse_bitstream(bs[: len(bs) // 2], n)            # truncated stream
    with pytest.raises(capi.IcspError):
        capi.parse_header(b"\x01" + bs[1:])                    # wrong magic (DEC:18)
    # a second geometry and all-intra
    w, h = 64, 48
    c2 = clipgen.synth_clip("tablelike", 3, width=w, height=h)
    o2 = po.encode_sequence(c2, w, h, 1, 1, 1)
    bs2 = capi.write_bitstream(w, h, 1, 1, 1, o2["levels"], o2["acflag"], o2["mpm"], o2["mvd"])
    _, g2 = capi.parse_bitstream(bs2, 3)
    assert np.array_equal(g2["levels"][:, :-1], o2["levels"][:, :-1]) and np.array_equal(g2["mpm"], o2["mpm"])


@pytest.mark.gpu
def test_gpu_decoder_matches_reference_decoder_hashes(golden_dir):
    """HIP decode path == the reference decoder binary's output (SHA-256 fixtures), PSNR line included."""
    for case in _cases(golden_dir):
        clip, o, bs = _stream(case)
        dec = capi.decode_bitstream(bs, case["nframes"])
        assert hashlib.sha256(dec.tobytes()).hexdigest() == case["decoded_sha256"], case
        assert abs(clipgen.psnr_y(clip, dec, W, H) - case["psnr"]) < 1e-4 + 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,q,period,w,h", [
    ("stefanlike", 5, 8, 3, 64, 48), ("mobilelike", 3, 1, 1, 416, 240), ("staticlike", 4, 16, 4, W, H),
    ("tablelike", 3, 16, 3, 1920, 1088), ("newslike", 2, 16, 2, 32, 16), ("footballlike", 24, 16, 6, W, H),
])
def test_gpu_decoder_matches_oracle_decoder(name, n, q, period, w, h):
    clip = clipgen.synth_clip(name, n, width=w, height=h)
    o = po.encode_sequence(clip, w, h, q, q, period, nthreads=4)
    bs = capi.write_bitstream(w, h, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    want = po.decode_bitstream(bs, n)
    got = capi.decode_bitstream(bs, n)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError(f"{len(bad)} decoded bytes differ, first at frame {bad[0][0]} offset {bad[0][1]}")


@pytest.mark.gpu
def test_gpu_encode_then_decode_in_place():
    """Encoder output left resident, decoded on the same context: equals the oracle decoder on the same syntax, and stays
    within a grey level or two of the encoder's own reconstruction."""
    n, q, period = 9, 16, 3
    clip = clipgen.synth_clip("coastguardlike", n)
    enc = capi.Encoder(W, H, q, q, period, max_frames=n)
    o = enc.encode(clip)
    enc.decode_resident(0, n)
    dec = enc.download(0, n, what=("recon",))["recon"]
    enc.close()
    want = po.decode_sequence(o["levels"], o["mpm"], o["mvd"], W, H, q, q, period)
    assert np.array_equal(dec, want)
    d = np.abs(dec.astype(int) - o["recon"].astype(int))
    assert d.max() <= 2 and (d > 0).mean() < 0.02
