"""CPU-only checks of the product's host side: the C-ABI library loads and exports every declared symbol, and the
host bit packer reproduces the reference's .bin byte for byte when fed the (oracle-computed) encoder outputs."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen
from oracle import pyoracle as po

W, H = 352, 288
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "icsp_hip.h")).read()
    declared = set(re.findall(r"\b(icsp_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.icsp_strerror(0) == b"success"
    assert lib.icsp_kernel_name(0) == b"k_intra_luma"


def _one_mb_stream(period, lv, ac, mpm, mvd):
    return capi.write_bitstream(32, 32, 16, 16, period, lv, ac, mpm, mvd)


def test_code_table_matches_reference(golden_dir):
    """Every magnitude class edge of the shared DC/AC/MV code (ENC:5417-5602)."""
    codes = json.load(open(os.path.join(golden_dir, "codes.json")))
    z_lv = np.zeros((1, 4, 6, 64), np.int16)
    z_ac = np.ones((1, 4, 6), np.uint8)
    z_mpm = np.zeros((1, 4, 4), np.uint8)
    z_mvd = np.zeros((1, 4, 2), np.int8)
    for v, bits in codes.items():
        v = int(v)
        # as a DC level of the first block of an all-intra stream: 2 mode bits, then the code
        lv = z_lv.copy()
        lv[0, 0, 0, 0] = v
        body = np.unpackbits(np.frombuffer(_one_mb_stream(0, lv, z_ac, z_mpm, z_mvd)[14:], np.uint8))
        assert "".join(map(str, body[2: 2 + len(bits)])) == bits, v
        if abs(v) <= 127:
            # as mvd.x of the first MB of a P frame (frame 1 of a period-2 stream)
            mvd = z_mvd.copy()
            mvd[0, 0, 0] = v
            bs = _one_mb_stream(2, np.concatenate([z_lv, z_lv]), np.concatenate([z_ac, z_ac]),
                                np.concatenate([z_mpm, z_mpm]), np.concatenate([z_mvd, mvd]))
            body = np.unpackbits(np.frombuffer(bs[14:], np.uint8))
            off = 4 * (4 * (2 + 2 + 1 + 63) + 2 * (2 + 1 + 63))      # the intra frame: 4 MBs
            assert "".join(map(str, body[off + 1: off + 1 + len(bits)])) == bits, v


def test_header_bytes():
    lv = np.zeros((1, 4, 6, 64), np.int16)
    ac = np.ones((1, 4, 6), np.uint8)
    bs = capi.write_bitstream(32, 32, 8, 16, 10, lv, ac, np.zeros((1, 4, 4), np.uint8), np.zeros((1, 4, 2), np.int8))
    assert bs[:5] == b"\x00ICSP" and bs[5:9] == bytes([32, 0, 32, 0]) and bs[9:12] == bytes([8, 16, 0])
    assert int.from_bytes(bs[12:14], "little") == 10 << 7


def test_small_bins_byte_identical(golden_dir):
    for fn, name, n, q, period in (("foremanlike_2f_q16_p0.bin", "foremanlike", 2, 16, 0),
                                   ("stefanlike_3f_q8_p3.bin", "stefanlike", 3, 8, 3)):
        clip = clipgen.synth_clip(name, n)
        o = po.encode_sequence(clip, W, H, q, q, period)
        bs = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
        assert bs == open(os.path.join(golden_dir, fn), "rb").read()


def test_stream_hashes(golden_dir):
    """All reference CLI runs recorded in streams.json up to 12 frames, plus BASELINE configs 1-3 at 300 frames."""
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    full = {("foremanlike", 300, 16, 0), ("stefanlike", 300, 8, 10)}
    n_checked = 0
    for s in streams:
        key = (s["clip"], s["nframes"], s["qp"], s["intra_period"])
        if "bin_sha256" not in s or (s["nframes"] > 12 and key not in full):
            continue
        clip = clipgen.synth_clip(s["clip"], s["nframes"])
        o = po.encode_sequence(clip, W, H, s["qp"], s["qp"], s["intra_period"], nthreads=8)
        bs = capi.write_bitstream(W, H, s["qp"], s["qp"], s["intra_period"], o["levels"], o["acflag"], o["mpm"], o["mvd"])
        assert len(bs) == s["bin_bytes"], key
        assert hashlib.sha256(bs).hexdigest() == s["bin_sha256"], key
        n_checked += 1
    assert n_checked >= 8


def test_create_without_device_fails_loudly_not_silently():
    """On a box without a GPU the product must refuse, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.IcspError, match="no usable HIP device"):
        capi.Encoder(W, H, 16, 16, 0, max_frames=1)
