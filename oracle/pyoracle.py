"""ctypes bindings for the parity oracle (oracle/libicsp_oracle.so) and, where it has been built,
the real reference (oracle/_ref/libicsp_ref.so, oracle/_ref/icsp_ref).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from icspcodec_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libicsp_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "libicsp_ref.so")
REF_ENC = os.path.join(REF_DIR, "icsp_ref")
REF_DEC = os.path.join(REF_DIR, "icsp_ref_dec")

_p = np.ctypeslib.ndpointer


def build(ref: bool = True) -> None:
    """(Re)build the oracle, and oracle/_ref when /root/reference is present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/source/encoder"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        _lib = C.CDLL(ORACLE_SO)
        _lib.icsp_oracle_irt2.restype = C.c_double
    return _lib


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.ref_irt2.restype = C.c_double
    return _ref


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------ block level (oracle and ref)
def costable(which="oracle"):
    out = np.zeros(64, np.float64)
    (lib().icsp_oracle_costable if which == "oracle" else ref().ref_costable)(_vp(out))
    return out.reshape(8, 8)


def irt2(which="oracle"):
    return lib().icsp_oracle_irt2() if which == "oracle" else ref().ref_irt2()


def dct8x8(blk, which="oracle"):
    i = _c(blk, np.int32).reshape(64)
    o = np.zeros(64, np.float64)
    fn = {"oracle": lambda: lib().icsp_oracle_dct8x8, "ref": lambda: ref().ref_dct_block,
          "refc": lambda: ref().ref_cdct_block}[which]()
    fn(_vp(i), _vp(o))
    return o.reshape(8, 8)


def idct8x8(blk, which="oracle"):
    i = _c(blk, np.int32).reshape(64)
    o = np.zeros(64, np.float64)
    fn = {"oracle": lambda: lib().icsp_oracle_idct8x8, "ref": lambda: ref().ref_idct_block,
          "refc": lambda: ref().ref_cidct_block}[which]()
    fn(_vp(i), _vp(o))
    return o.reshape(8, 8)


def quant(coef, qdc, qac, chroma=False, which="oracle"):
    """Returns (q[8,8], zigzag[64], iq[8,8], acflag)."""
    c = _c(coef, np.float64).reshape(64)
    if which == "ref":
        q = np.zeros(64, np.int32); zz = np.zeros(64, np.int32); iq = np.zeros(64, np.int32)
        ac = C.c_int(0)
        (ref().ref_quant_chroma if chroma else ref().ref_quant_luma)(_vp(c), qdc, qac, _vp(q), _vp(zz), _vp(iq), C.byref(ac))
        return q.reshape(8, 8), zz, iq.reshape(8, 8), ac.value
    L = lib()
    L.icsp_oracle_quant_luma.argtypes = [C.c_double, C.c_int]
    L.icsp_oracle_quant_chroma.argtypes = [C.c_double, C.c_int]
    f = L.icsp_oracle_quant_chroma if chroma else L.icsp_oracle_quant_luma
    q = np.array([f(float(c[i]), qdc if i == 0 else qac) for i in range(64)], np.int32)
    zz = np.zeros(64, np.int32)
    L.icsp_oracle_zigzag(_vp(q), _vp(zz))
    iq = q * np.where(np.arange(64) == 0, qdc, qac).astype(np.int32)
    return q.reshape(8, 8), zz, iq.reshape(8, 8), int(not np.any(q[1:]))


def pad(plane, padlen, which="oracle"):
    src = _c(plane, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * padlen, w + 2 * padlen), np.uint8)
    (lib().icsp_oracle_pad if which == "oracle" else ref().ref_pad)(_vp(src), _vp(dst), padlen, w, h)
    return dst


def sad16(cur, refblk, which="oracle"):
    a = _c(cur, np.uint8).reshape(256); b = _c(refblk, np.uint8).reshape(256)
    if which == "oracle":
        return lib().icsp_oracle_sad16(_vp(a), 16, _vp(b), 16)
    return ref().ref_sad16(_vp(a), _vp(b))


def me_walk(state):
    dx = np.zeros(64, np.int32); dy = np.zeros(64, np.int32)
    lib().icsp_oracle_me_walk(int(state), _vp(dx), _vp(dy))
    return dx, dy


def me_frame(curY, prevY, which="oracle"):
    cur = _c(curY, np.uint8); prev = _c(prevY, np.uint8)
    h, w = cur.shape
    nmb = (w // 16) * (h // 16)
    mx = np.zeros(nmb, np.int32); my = np.zeros(nmb, np.int32); ns = np.zeros(nmb, np.int32)
    if which == "oracle":
        lib().icsp_oracle_me_frame(_vp(cur), _vp(prev), w, h, _vp(mx), _vp(my), _vp(ns))
    else:
        ref().ref_me_frame(_vp(cur), _vp(prev), w, h, _vp(mx), _vp(my))
    return mx, my, ns


# ------------------------------------------------------------------ frame / sequence level
def _alloc(nframes, w, h):
    nmb = (w // 16) * (h // 16)
    return dict(
        levels=np.zeros((nframes, nmb, 6, 64), np.int16),
        acflag=np.zeros((nframes, nmb, 6), np.uint8),
        mpm=np.zeros((nframes, nmb, 4), np.uint8),
        mvd=np.zeros((nframes, nmb, 2), np.int8),
        recon=np.zeros((nframes, w * h * 3 // 2), np.uint8),
    )


def intra_frame(frame, w, h, qdc, qac, want_dbg=False):
    f = _c(frame, np.uint8)
    o = _alloc(1, w, h)
    nmb = (w // 16) * (h // 16)
    coef = np.zeros((nmb, 6, 64), np.float64) if want_dbg else None
    mode = np.zeros((nmb, 4), np.uint8) if want_dbg else None
    lib().icsp_oracle_intra_frame(_vp(f), w, h, qdc, qac, _vp(o["levels"]), _vp(o["acflag"]), _vp(o["mpm"]),
                                  _vp(o["recon"]), _vp(coef), _vp(mode))
    r = {k: v[0] for k, v in o.items()}
    r["coef"], r["mode"] = coef, mode
    return r


def inter_frame(frame, prev_recon, w, h, qdc, qac, want_dbg=False):
    f = _c(frame, np.uint8); p = _c(prev_recon, np.uint8)
    o = _alloc(1, w, h)
    nmb = (w // 16) * (h // 16)
    coef = np.zeros((nmb, 6, 64), np.float64) if want_dbg else None
    mv = np.zeros((nmb, 2), np.int8) if want_dbg else None
    lib().icsp_oracle_inter_frame(_vp(f), _vp(p), w, h, qdc, qac, _vp(o["levels"]), _vp(o["acflag"]), _vp(o["mvd"]),
                                  _vp(o["recon"]), _vp(coef), _vp(mv))
    r = {k: v[0] for k, v in o.items()}
    r["coef"], r["mv"] = coef, mv
    return r


def encode_sequence(yuv, w, h, qdc, qac, intra_period, nthreads=1):
    y = _c(yuv, np.uint8)
    n = y.shape[0]
    o = _alloc(n, w, h)
    rc = lib().icsp_oracle_encode_sequence(_vp(y), n, w, h, qdc, qac, intra_period, nthreads, _vp(o["levels"]),
                                           _vp(o["acflag"]), _vp(o["mpm"]), _vp(o["mvd"]), _vp(o["recon"]))
    assert rc == 0
    return o


def ref_encode_frames(yuv, w, h, qdc, qac, intra_period):
    """The real reference's frame encoders on in-memory frames (needs oracle/_ref)."""
    y = _c(yuv, np.uint8)
    n = y.shape[0]
    o = _alloc(n, w, h)
    nmb = (w // 16) * (h // 16)
    o["mode"] = np.zeros((n, nmb, 4), np.uint8)
    o["mv"] = np.zeros((n, nmb, 2), np.int8)
    rc = ref().ref_encode_frames(_vp(y), n, w, h, qdc, qac, intra_period, _vp(o["levels"]), _vp(o["acflag"]),
                                 _vp(o["mpm"]), _vp(o["mvd"]), _vp(o["recon"]), _vp(o["mode"]), _vp(o["mv"]))
    assert rc == 0, rc
    return o


def ref_code_value(v):
    bits = np.zeros(32, np.uint8)
    n = ref().ref_code_value(int(v), _vp(bits))
    return bits[:n].copy()


def run_ref_encoder(yuv_path, nframes, qp, intra_period, threads=0, cwd=None, qpdc=None, qpac=None):
    """Run the reference CLI binary (oracle/_ref/icsp_ref).  Returns CompletedProcess."""
    cmd = [REF_ENC, "-i", os.path.basename(yuv_path), "-n", str(nframes)]
    if qpdc is not None:
        cmd += ["--qpdc", str(qpdc), "--qpac", str(qpac)]
    else:
        cmd += ["-q", str(qp)]
    cmd += ["--intraPeriod", str(intra_period)]
    if threads:
        cmd += ["--EnMultiThread", str(threads)]
    return subprocess.run(cmd, cwd=cwd or os.path.dirname(yuv_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


# ------------------------------------------------------------------ decoder restatement (icsp_oracle_dec.c)
def parse_header(bs: bytes):
    """(w, h, qdc, qac, period) of a .bin image (DEC:14-37)."""
    b = np.frombuffer(bs, np.uint8)
    v = [C.c_int(0) for _ in range(5)]
    L = lib()
    L.icsp_oracle_parse_header.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 5
    if L.icsp_oracle_parse_header(_vp(b), b.size, *[C.byref(x) for x in v]):
        raise ValueError("not an ICSP stream")
    return tuple(x.value for x in v)


def parse_bitstream(bs: bytes, nframes: int):
    """readBlockData (DEC:38-405): the .bin image -> dict(levels, acflag, mpm, mvd) + header fields."""
    w, h, qdc, qac, period = parse_header(bs)
    b = np.frombuffer(bs, np.uint8)
    o = _alloc(nframes, w, h)
    del o["recon"]
    L = lib()
    L.icsp_oracle_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4
    if L.icsp_oracle_parse(_vp(b), b.size, nframes, _vp(o["levels"]), _vp(o["acflag"]), _vp(o["mpm"]), _vp(o["mvd"])):
        raise ValueError("stream ends early")
    o.update(width=w, height=h, qdc=qdc, qac=qac, period=period)
    return o


def decode_sequence(levels, mpm, mvd, w, h, qdc, qac, period):
    """IcspCodec::decoding (DEC.h:286-312) on parsed arrays -> uint8 [n][W*H*3/2] (what check_test_*_yuv.yuv holds)."""
    lv = _c(levels, np.int16); mp = _c(mpm, np.uint8); mv = _c(mvd, np.int8)
    n = lv.shape[0]
    out = np.zeros((n, w * h * 3 // 2), np.uint8)
    lib().icsp_oracle_decode_sequence(_vp(lv), _vp(mp), _vp(mv), n, w, h, qdc, qac, period, _vp(out))
    return out


def decode_bitstream(bs: bytes, nframes: int):
    p = parse_bitstream(bs, nframes)
    return decode_sequence(p["levels"], p["mpm"], p["mvd"], p["width"], p["height"], p["qdc"], p["qac"], p["period"])


def dec_idct8x8(blk):
    b = _c(np.asarray(blk).reshape(64), np.int32)
    out = np.zeros(64, np.float64)
    lib().icsp_oracle_dec_idct8x8(_vp(b), _vp(out))
    return out.reshape(8, 8)


def psnr_y(orig, dec, w, h):
    a = _c(orig, np.uint8); d = _c(dec, np.uint8)
    L = lib()
    L.icsp_oracle_psnr_y.restype = C.c_double
    return L.icsp_oracle_psnr_y(_vp(a), _vp(d), a.shape[0], w, h)


def run_ref_decoder(bs: bytes, clip, nframes, qdc, qac, period, workdir):
    """Run the reference decoder binary (oracle/_ref/icsp_ref_dec) on a .bin image.  Returns (decoded uint8 [n][fsz], psnr).
    It opens files literally named output\\<bin> and data\\<yuv> in its working directory (DEC.h:241, 323)."""
    import glob
    import re
    binname, yuvname = "t_compCIF.bin", "t_cif.yuv"
    open(os.path.join(workdir, "output\\" + binname), "wb").write(bs)
    np.asarray(clip, np.uint8).tofile(os.path.join(workdir, "data\\" + yuvname))
    for f in glob.glob(os.path.join(workdir, "check_test_*")) + glob.glob(os.path.join(workdir, "experimental_Result_Decoding.txt")):
        os.remove(f)
    r = subprocess.run([REF_DEC, str(nframes), binname, str(qdc), str(qac), str(period), yuvname], cwd=workdir,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(r.stdout.decode(errors="replace"))
    out = glob.glob(os.path.join(workdir, "check_test_*yuv.yuv"))
    dec = np.fromfile(out[0], np.uint8).reshape(nframes, -1)
    line = open(os.path.join(workdir, "experimental_Result_Decoding.txt")).read()
    return dec, float(re.search(r"PSNR: ([0-9.]+)", line).group(1))
