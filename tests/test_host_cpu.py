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
    # ... and nothing else: the launch layer between the two translation units (icsp_kernels.h) and the host helpers stay inside
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in ("T", "t")}
    assert exported == declared, exported ^ declared


def test_poisoned_context_keeps_returning_hip_error():
    """A failed launch-path call (kernel launch, event record, cross-stream wait) poisons its context: every later entry point
    answers ICSP_ERR_HIP instead of running on with an ordering edge missing.  The hook builds a context in that state; no
    call below may reach the HIP runtime (there is no device here)."""
    import ctypes as C
    lib = capi.load()
    ctx = C.c_void_p()
    assert lib.icsp_debug_poisoned_context(C.byref(ctx)) == 0 and ctx
    ERR_HIP = 5
    buf = np.zeros(352 * 288 * 3 // 2, np.uint8)
    n64 = C.c_uint64(0)
    ms, cnt = C.c_double(0), C.c_longlong(0)
    view = capi.DeviceView()
    vp = buf.ctypes.data_as(C.c_void_p)
    calls = {
        "icsp_sync": lambda: lib.icsp_sync(ctx),
        "icsp_upload": lambda: lib.icsp_upload(ctx, vp, 0, 1),
        "icsp_upload_sync": lambda: lib.icsp_upload_sync(ctx, vp, 0, 1),
        "icsp_encode_resident": lambda: lib.icsp_encode_resident(ctx, 0, 1),
        "icsp_encode_resident_many": lambda: lib.icsp_encode_resident_many(ctx, 1, (C.c_int * 1)(0), (C.c_int * 1)(1)),
        "icsp_encode_gop": lambda: lib.icsp_encode_gop(ctx, vp, 1, None, None, None, None, None),
        "icsp_encode_gop_packed": lambda: lib.icsp_encode_gop_packed(ctx, vp, 1, None, vp, buf.size, C.byref(n64)),
        "icsp_download": lambda: lib.icsp_download(ctx, 0, 1, None, None, None, None, vp),
        "icsp_download_debug": lambda: lib.icsp_download_debug(ctx, 0, 1, None, None),
        "icsp_decode_resident": lambda: lib.icsp_decode_resident(ctx, 0, 1),
        "icsp_upload_syntax": lambda: lib.icsp_upload_syntax(ctx, 0, 1, vp, vp, vp),
        "icsp_pack_count": lambda: lib.icsp_pack_count(ctx, 0, 1, C.byref(n64)),
        "icsp_pack_into": lambda: lib.icsp_pack_into(ctx, 0, 1, 0, vp, buf.size),
        "icsp_pack_bits": lambda: lib.icsp_pack_bits(ctx, 0, 1, vp, buf.size, C.byref(n64)),
        "icsp_prepare": lambda: lib.icsp_prepare(ctx),
        "icsp_host_warm": lambda: lib.icsp_host_warm(ctx, vp, 4096),
        "icsp_copy_streams": lambda: lib.icsp_copy_streams(ctx, 1),
        "icsp_set_groups": lambda: lib.icsp_set_groups(ctx, 1, 1),
        "icsp_single_stream": lambda: lib.icsp_single_stream(ctx, 1),
        "icsp_debug_last_choice": lambda: lib.icsp_debug_last_choice(ctx, None, None, None, None, None, None),
        "icsp_device_view": lambda: lib.icsp_device_view(ctx, C.byref(view)),
        "icsp_debug_keep_coef": lambda: lib.icsp_debug_keep_coef(ctx, 1),
        "icsp_download_coef": lambda: lib.icsp_download_coef(ctx, 0, 1, vp),
        "icsp_profile_enable": lambda: lib.icsp_profile_enable(ctx, 1),
        "icsp_profile_reset": lambda: lib.icsp_profile_reset(ctx),
        "icsp_profile_get": lambda: lib.icsp_profile_get(ctx, 0, C.byref(ms), C.byref(cnt)),
    }
    for rnd in range(2):                                   # sticky: the second round answers the same
        for name, f in calls.items():
            assert f() == ERR_HIP, (name, rnd)
    assert b"icsp_debug_poisoned_context" in lib.icsp_last_error(ctx)
    assert lib.icsp_destroy(ctx) == 0


def test_build_flags_turn_unchecked_hip_calls_into_errors():
    """Every HIP return code is checked or explicitly discarded: the product build carries -Werror=unused-value (hipError_t is
    [[nodiscard]])."""
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert '"-Werror=unused-value"' in src and "-Wno-unused-value" not in src


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


def body_bits(levels, acflag, mvd, period):
    """Bit count of the stream body from the syntax (ENC:4923-5236) and the value code's lengths (ENC:5417-5602)."""
    def vlen(v):
        a = np.abs(v.astype(np.int64))
        e = np.floor(np.log2(np.maximum(a, 1))).astype(np.int64)
        return np.where(a == 0, 2, np.where(a == 1, 4, np.where(a <= 31, 4 + e, 2 * np.minimum(e, 11))))
    n = levels.shape[0]
    intra = np.array([period == 0 or f % period == 0 for f in range(n)])
    dc = vlen(levels[..., 0]) + 1
    ac = np.where(acflag == 1, 63, vlen(levels[..., 1:]).sum(-1))
    total = int((dc + ac).sum())
    total += int(intra.sum()) * levels.shape[1] * 8                                  # 4 x (MPMFlag + intraPredMode)
    total += int((1 + vlen(mvd[~intra]).sum(-1)).sum())                              # mode flag + mvd x, y
    return total


def test_assemble_pieces_equals_one_pass_writer():
    """icsp_bitstream_assemble: per-GOP bodies concatenated bit-wise == the sequential writer on the whole sequence."""
    W, H, q, period, n = 64, 48, 8, 3, 8                                             # GOPs of 3, 3, 2 frames
    clip = clipgen.synth_clip("stefanlike", n, width=W, height=H)
    o = po.encode_sequence(clip, W, H, q, q, period)
    whole = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    pieces = []
    for g0 in range(0, n, period):
        s = slice(g0, min(g0 + period, n))
        nb = body_bits(o["levels"][s], o["acflag"][s], o["mvd"][s], period)
        b = bytearray(capi.write_bitstream(W, H, q, q, period, o["levels"][s], o["acflag"][s], o["mpm"][s], o["mvd"][s])[14:])
        assert len(b) == nb // 8 + 1
        if nb % 8:
            b[-1] = (b[-1] << (8 - nb % 8)) & 0xff                                   # undo the right-aligned final byte
        pieces.append((bytes(b[: (nb + 7) // 8]), nb))
    assert sum(nb for _, nb in pieces) // 8 + 1 + 14 == len(whole)
    assert capi.assemble_bitstream(W, H, q, q, period, pieces) == whole
    assert capi.assemble_bitstream(W, H, q, q, period, []) == whole[:14] + b"\x00"


def test_assemble_random_splits_and_parser_on_garbage():
    """Bit-wise concatenation at arbitrary bit boundaries; the parser returns a status on garbage instead of crashing."""
    rng = np.random.default_rng(7)
    W, H, q, period, n = 64, 48, 8, 3, 6
    clip = clipgen.synth_clip("tablelike", n, width=W, height=H)
    o = po.encode_sequence(clip, W, H, q, q, period)
    whole = capi.write_bitstream(W, H, q, q, period, o["levels"], o["acflag"], o["mpm"], o["mvd"])
    nb = body_bits(o["levels"], o["acflag"], o["mvd"], period)
    body = bytearray(whole[14:])
    if nb % 8:
        body[-1] = (body[-1] << (8 - nb % 8)) & 0xff
    bits = np.unpackbits(np.frombuffer(bytes(body), np.uint8))[:nb]
    for _ in range(20):
        cuts = sorted(set([0, nb] + list(rng.integers(0, nb, size=int(rng.integers(0, 6))))))
        pieces = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            seg = bits[a:b]
            pieces.append((np.packbits(seg).tobytes(), len(seg)))          # zero-padded, MSB first
        assert capi.assemble_bitstream(W, H, q, q, period, pieces) == whole
    # garbage after a valid header: ICSP_OK or ICSP_ERR_RANGE, never a crash; a bad header is rejected
    for _ in range(30):
        junk = whole[:14] + rng.integers(0, 256, size=int(rng.integers(0, 4000)), dtype=np.uint8).tobytes()
        try:
            capi.parse_bitstream(junk, n)
        except capi.IcspError:
            pass
    with pytest.raises(capi.IcspError):
        capi.parse_bitstream(b"\x00ICS", 1)


def test_isa_guard_on_the_shipped_compile():
    """build() disassembles the compile that ships (not only the -DICSP_NO_FMA one): fused multiply-adds per kernel must
    equal the recorded table, the decoder kernels must hold none, and no kernel may spill to scratch."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    table = json.load(open(g.FMA_TABLE))
    assert table and all("k_dec_" not in k for k in table)
    meta = "- .agpr_count: 0\n    .name: {n}\n    .private_segment_fixed_size: {s}\n    .wavefront_size: 64\n"
    kern = next(iter(table))
    ok = "".join(f"{k}:\n" + "\tv_fmac_f64_e32 v[0:1], v[2:3], v[4:5]\n" * v + "\ts_endpgm\n" for k, v in table.items()) + \
         "_ZN1xk_dec_blocksE:\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n\ts_endpgm\n" + "".join(meta.format(n=k, s=0) for k in table)
    assert g.check_device_asm(ok)[kern] == table[kern]
    with pytest.raises(RuntimeError, match="must be none"):
        g.check_device_asm(ok.replace("v_add_f64 v[0:1], v[0:1], v[2:3]", "v_fma_f64 v[0:1], v[0:1], v[2:3], v[4:5]"))
    with pytest.raises(RuntimeError, match="counts moved"):
        g.check_device_asm(ok.replace(f"{kern}:\n", f"{kern}:\n\tv_fma_f64 v[0:1], v[0:1], v[2:3], v[4:5]\n", 1))
    with pytest.raises(RuntimeError, match="scratch"):
        g.check_device_asm(ok.replace(".private_segment_fixed_size: 0", ".private_segment_fixed_size: 28", 1))
