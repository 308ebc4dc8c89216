"""ctypes binding of the C ABI in include/icsp_hip.h (icspcodec_amd/libicsp_hip.so).

This is the host-side mirror used by tests and bench.py; the production host is the C++ program
icspcodec_amd/csrc/icsp_enc_main.cpp which links the same library.  There is no CPU fallback: if the HIP library
is missing or no device is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICSP_LIB") or os.path.join(HERE, "libicsp_hip.so")      # ICSP_LIB: another build of the same ABI (A/B experiments)

# every symbol include/icsp_hip.h declares
SYMBOLS = [
    "icsp_create", "icsp_destroy", "icsp_strerror", "icsp_device_count", "icsp_last_error", "icsp_encode_gop", "icsp_encode_gop_packed", "icsp_upload",
    "icsp_encode_resident", "icsp_encode_resident_many", "icsp_sync", "icsp_download", "icsp_device_view", "icsp_download_debug",
    "icsp_debug_keep_coef", "icsp_download_coef", "icsp_profile_enable", "icsp_profile_reset", "icsp_profile_get",
    "icsp_kernel_name", "icsp_bitstream_bound", "icsp_write_bitstream", "icsp_pack_bits", "icsp_bitstream_assemble",
    "icsp_parse_header", "icsp_parse_bitstream", "icsp_upload_syntax", "icsp_decode_resident",
    "icsp_bitstream_begin", "icsp_bitstream_header", "icsp_pack_count", "icsp_pack_into", "icsp_prepare", "icsp_bitstream_place", "icsp_bitstream_end", "icsp_host_alloc", "icsp_host_free", "icsp_host_register", "icsp_host_unregister", "icsp_host_warm", "icsp_copy_streams", "icsp_upload_sync",
    "icsp_set_groups", "icsp_single_stream", "icsp_debug_poisoned_context", "icsp_debug_last_choice", "icsp_debug_stream_pool", "icsp_debug_plan_turns",
    "icsp_device_pci_bus_id", "icsp_numa_node_of_pci", "icsp_device_numa_node", "icsp_numa_nodes", "icsp_numa_cpus", "icsp_parse_cpulist",
    "icsp_bind_thread_to_node", "icsp_populate_here", "icsp_chunk_device",
]
KERNELS = ["k_intra_luma", "k_chroma_dc", "k_residual", "k_me", "k_frame_serial", "k_pack", "k_decode"]


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("qp_dc", C.c_int), ("qp_ac", C.c_int), ("intra_period", C.c_int)]


class DeviceView(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("levels", C.c_void_p), ("acflag", C.c_void_p), ("mpm_mode", C.c_void_p),
                ("mvd", C.c_void_p), ("recon", C.c_void_p), ("stream", C.c_void_p), ("max_frames", C.c_int), ("n_mb", C.c_int)]


class IcspError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load libicsp_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IcspError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        lib.icsp_strerror.restype = C.c_char_p
        lib.icsp_last_error.restype = C.c_char_p
        lib.icsp_last_error.argtypes = [C.c_void_p]
        lib.icsp_kernel_name.restype = C.c_char_p
        lib.icsp_bitstream_bound.restype = C.c_size_t
        lib.icsp_bitstream_bound.argtypes = [C.POINTER(Params), C.c_int]
        vp = C.c_void_p
        lib.icsp_create.argtypes = [C.POINTER(vp), C.POINTER(Params), C.c_int, C.c_int]
        lib.icsp_destroy.argtypes = [vp]
        lib.icsp_encode_gop.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
        lib.icsp_encode_gop_packed.argtypes = [vp, vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]
        lib.icsp_upload.argtypes = [vp, vp, C.c_int, C.c_int]
        lib.icsp_encode_resident.argtypes = [vp, C.c_int, C.c_int]
        if hasattr(lib, "icsp_encode_resident_many"):         # (absent from earlier rounds' builds, which ICSP_LIB may name)
            lib.icsp_encode_resident_many.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.icsp_sync.argtypes = [vp]
        lib.icsp_download.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        lib.icsp_device_view.argtypes = [vp, C.POINTER(DeviceView)]
        lib.icsp_download_debug.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        lib.icsp_debug_keep_coef.argtypes = [vp, C.c_int]
        lib.icsp_download_coef.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.icsp_profile_enable.argtypes = [vp, C.c_int]
        lib.icsp_profile_reset.argtypes = [vp]
        lib.icsp_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        lib.icsp_write_bitstream.argtypes = [C.POINTER(Params), C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.icsp_pack_bits.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_uint64)]
        lib.icsp_pack_count.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        lib.icsp_pack_into.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, vp, C.c_size_t]
        lib.icsp_prepare.argtypes = [vp]
        lib.icsp_bitstream_header.argtypes = [C.POINTER(Params), C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.icsp_bitstream_end.argtypes = [vp, C.c_uint64]
        lib.icsp_host_register.argtypes = [vp, C.c_size_t, C.c_int]
        lib.icsp_host_unregister.argtypes = [vp]
        lib.icsp_host_warm.argtypes = [vp, vp, C.c_size_t]
        lib.icsp_copy_streams.argtypes = [vp, C.c_int]
        lib.icsp_upload_sync.argtypes = [vp, vp, C.c_int, C.c_int]
        lib.icsp_bitstream_assemble.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64), vp, C.c_size_t,
                                                C.POINTER(C.c_size_t)]
        lib.icsp_parse_header.argtypes = [vp, C.c_size_t, C.POINTER(Params)]
        lib.icsp_parse_bitstream.argtypes = [vp, C.c_size_t, C.c_int, vp, vp, vp, vp]
        lib.icsp_upload_syntax.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
        lib.icsp_decode_resident.argtypes = [vp, C.c_int, C.c_int]
        lib.icsp_set_groups.argtypes = [vp, C.c_int, C.c_int]
        lib.icsp_single_stream.argtypes = [vp, C.c_int]
        lib.icsp_debug_poisoned_context.argtypes = [C.POINTER(vp)]
        _lib = lib
    return _lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def write_bitstream(width, height, qp_dc, qp_ac, intra_period, levels, acflag, mpm, mvd) -> bytes:
    """Host back end: the reference's makebitstream (ENC:4849-6334) on the encoder outputs.  Needs no GPU."""
    lib = load()
    p = Params(width, height, qp_dc, qp_ac, intra_period)
    lv = np.ascontiguousarray(levels, np.int16); ac = np.ascontiguousarray(acflag, np.uint8)
    mp = np.ascontiguousarray(mpm, np.uint8); mv = np.ascontiguousarray(mvd, np.int8)
    n = lv.shape[0]
    cap = lib.icsp_bitstream_bound(C.byref(p), n)
    out = np.zeros(cap, np.uint8)
    nbytes = C.c_size_t(0)
    rc = lib.icsp_write_bitstream(C.byref(p), n, _vp(lv), _vp(ac), _vp(mp), _vp(mv), _vp(out), cap, C.byref(nbytes))
    if rc:
        raise IcspError(lib.icsp_strerror(rc).decode())
    return out[: nbytes.value].tobytes()


def assemble_bitstream(width, height, qp_dc, qp_ac, intra_period, pieces) -> bytes:
    """Host: header + bit-wise concatenation of `pieces` = [(bytes-like body, nbits), ...] in frame order + the reference's
    final byte -> the .bin image (icsp_bitstream_assemble).  Needs no GPU."""
    lib = load()
    p = Params(width, height, qp_dc, qp_ac, intra_period)
    bufs = [np.frombuffer(bytes(b), np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, np.uint8) for b, _ in pieces]
    k = len(pieces)
    ptrs = (C.c_void_p * max(k, 1))(*[b.ctypes.data for b in bufs])
    bits = (C.c_uint64 * max(k, 1))(*[int(nb) for _, nb in pieces])
    cap = 14 + sum(int(nb) for _, nb in pieces) // 8 + 3
    out = np.zeros(cap, np.uint8)
    nbytes = C.c_size_t(0)
    rc = lib.icsp_bitstream_assemble(C.byref(p), k, ptrs, bits, _vp(out), cap, C.byref(nbytes))
    if rc:
        raise IcspError(lib.icsp_strerror(rc).decode())
    return out[: nbytes.value].tobytes()


def finish_image(width, height, qp_dc, qp_ac, intra_period, image: np.ndarray, total_bits: int) -> bytes:
    """Header + the reference's final byte on an image whose body icsp_pack_into has filled (icsp_bitstream_header/_end)."""
    lib = load()
    p = Params(width, height, qp_dc, qp_ac, intra_period)
    n = C.c_size_t(0)
    rc = lib.icsp_bitstream_header(C.byref(p), total_bits, _vp(image), image.size, C.byref(n))
    if rc == 0:
        rc = lib.icsp_bitstream_end(_vp(image), total_bits)
    if rc:
        raise RuntimeError(f"icsp_bitstream_header/end: {lib.icsp_strerror(rc).decode()}")
    return image[: n.value].tobytes()


def host_alloc_array(shape, dtype) -> np.ndarray:
    """A numpy array in pinned host memory (icsp_host_alloc); release it with host_free_array."""
    lib = load()
    lib.icsp_host_alloc.restype = C.c_void_p
    lib.icsp_host_alloc.argtypes = [C.c_size_t]
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.icsp_host_alloc(max(nbytes, 1))
    if not p:
        raise IcspError("icsp_host_alloc failed")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(p)
    a = np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)
    _pinned[a.ctypes.data] = p
    return a


def host_free_array(a: np.ndarray):
    lib = load()
    lib.icsp_host_free.argtypes = [C.c_void_p]
    p = _pinned.pop(a.ctypes.data, None)
    if p:
        lib.icsp_host_free(p)


_pinned = {}


def host_register(a: np.ndarray, read_only=False) -> bool:
    """Pins the array's memory for DMA (icsp_host_register); False when the runtime refuses."""
    return load().icsp_host_register(_vp(a), a.nbytes, 1 if read_only else 0) == 0


def host_unregister(a: np.ndarray) -> bool:
    return load().icsp_host_unregister(_vp(a)) == 0


def parse_header(bs: bytes) -> Params:
    """readHeader (DEC:14-37).  Needs no GPU."""
    lib = load()
    b = np.frombuffer(bs, np.uint8)
    p = Params()
    rc = lib.icsp_parse_header(_vp(b), b.size, C.byref(p))
    if rc:
        raise IcspError(lib.icsp_strerror(rc).decode())
    return p


def parse_bitstream(bs: bytes, nframes: int):
    """Host parser, the inverse of write_bitstream (readBlockData, DEC:38-405): (Params, dict of syntax arrays).  Needs no GPU."""
    lib = load()
    p = parse_header(bs)
    b = np.frombuffer(bs, np.uint8)
    nmb = (p.width // 16) * (p.height // 16)
    o = dict(levels=np.zeros((nframes, nmb, 6, 64), np.int16), acflag=np.zeros((nframes, nmb, 6), np.uint8),
             mpm=np.zeros((nframes, nmb, 4), np.uint8), mvd=np.zeros((nframes, nmb, 2), np.int8))
    rc = lib.icsp_parse_bitstream(_vp(b), b.size, nframes, _vp(o["levels"]), _vp(o["acflag"]), _vp(o["mpm"]), _vp(o["mvd"]))
    if rc:
        raise IcspError(lib.icsp_strerror(rc).decode())
    return p, o


def decode_bitstream(bs: bytes, nframes: int, device=0):
    """.bin image -> decoded frames uint8 [n][W*H*3/2] (what the reference decoder writes to check_test_*_yuv.yuv):
    host parse, device reconstruction."""
    p, o = parse_bitstream(bs, nframes)
    dec = Encoder(p.width, p.height, p.qp_dc, p.qp_ac, p.intra_period, device=device, max_frames=nframes)
    try:
        dec.upload_syntax(0, o["levels"], o["mpm"], o["mvd"])
        dec.decode_resident(0, nframes)
        return dec.download(0, nframes, what=("recon",))["recon"]
    finally:
        dec.close()


class Encoder:
    """One context on one device (icsp_create ... icsp_destroy)."""

    def __init__(self, width=352, height=288, qp_dc=16, qp_ac=16, intra_period=0, device=0, max_frames=300):
        self.lib = load()
        self.params = Params(width, height, qp_dc, qp_ac, intra_period)
        self.width, self.height = width, height
        self.nmb = (width // 16) * (height // 16)
        self.fsz = width * height * 3 // 2
        self.max_frames = max_frames
        self.ctx = C.c_void_p()
        rc = self.lib.icsp_create(C.byref(self.ctx), C.byref(self.params), device, max_frames)
        if rc:
            raise IcspError(f"icsp_create: {self.lib.icsp_strerror(rc).decode()}")

    def _chk(self, rc, what):
        if rc:
            raise IcspError(f"{what}: {self.lib.icsp_strerror(rc).decode()} [{self.lib.icsp_last_error(self.ctx).decode()}]")

    def close(self):
        if self.ctx:
            self.lib.icsp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, n):
        return dict(levels=np.zeros((n, self.nmb, 6, 64), np.int16), acflag=np.zeros((n, self.nmb, 6), np.uint8),
                    mpm=np.zeros((n, self.nmb, 4), np.uint8), mvd=np.zeros((n, self.nmb, 2), np.int8),
                    recon=np.zeros((n, self.fsz), np.uint8))

    def encode(self, yuv: np.ndarray, out: dict | None = None) -> dict:
        """icsp_encode_gop: host frames in, host results out (into `out`, a dict like the one returned, when given: arrays
        the caller keeps, possibly pinned)."""
        y = yuv if (isinstance(yuv, np.ndarray) and yuv.dtype == np.uint8 and yuv.flags.c_contiguous) else np.ascontiguousarray(yuv, np.uint8)
        y = y.reshape(-1, self.fsz)
        n = y.shape[0]
        o = out if out is not None else self._alloc(n)
        self._chk(self.lib.icsp_encode_gop(self.ctx, _vp(y), n, _vp(o["levels"]), _vp(o["acflag"]), _vp(o["mpm"]),
                                           _vp(o["mvd"]), _vp(o["recon"])), "icsp_encode_gop")
        return o

    def encode_packed(self, yuv: np.ndarray, recon: np.ndarray | None = None, body: np.ndarray | None = None):
        """icsp_encode_gop_packed: host frames in; the packed body (and the reconstruction, if an array is given) out.
        Returns (body bytes as uint8 array, bits)."""
        y = np.ascontiguousarray(yuv, np.uint8).reshape(-1, self.fsz)
        n = y.shape[0]
        if body is None:
            body = np.empty(self.lib.icsp_bitstream_bound(C.byref(self.params), n), np.uint8)
        bits = C.c_uint64(0)
        self._chk(self.lib.icsp_encode_gop_packed(self.ctx, _vp(y), n, _vp(recon) if recon is not None else None, _vp(body), body.nbytes,
                                                  C.byref(bits)), "icsp_encode_gop_packed")
        return body[: (bits.value + 7) // 8], bits.value

    def upload(self, yuv: np.ndarray, first=0):
        y = np.ascontiguousarray(yuv, np.uint8).reshape(-1, self.fsz)
        self._chk(self.lib.icsp_upload(self.ctx, _vp(y), first, y.shape[0]), "icsp_upload")
        self._chk(self.lib.icsp_sync(self.ctx), "icsp_sync")
        return y.shape[0]

    def encode_resident(self, first, n):
        self._chk(self.lib.icsp_encode_resident(self.ctx, first, n), "icsp_encode_resident")

    def encode_resident_many(self, ranges):
        """icsp_encode_resident_many: `ranges` = [(first, n), ...], disjoint, encoded as one batch."""
        k = len(ranges)
        f = (C.c_int * max(k, 1))(*[int(a) for a, _ in ranges])
        n = (C.c_int * max(k, 1))(*[int(b) for _, b in ranges])
        self._chk(self.lib.icsp_encode_resident_many(self.ctx, k, f, n), "icsp_encode_resident_many")

    def sync(self):
        self._chk(self.lib.icsp_sync(self.ctx), "icsp_sync")

    def download(self, first, n, what=("levels", "acflag", "mpm", "mvd", "recon")) -> dict:
        o = self._alloc(n)
        args = [_vp(o[k]) if k in what else None for k in ("levels", "acflag", "mpm", "mvd", "recon")]
        self._chk(self.lib.icsp_download(self.ctx, first, n, *args), "icsp_download")
        return {k: o[k] for k in what}

    def download_debug(self, first, n):
        mv = np.zeros((n, self.nmb, 2), np.int8); mode = np.zeros((n, self.nmb, 4), np.uint8)
        self._chk(self.lib.icsp_download_debug(self.ctx, first, n, _vp(mv), _vp(mode)), "icsp_download_debug")
        return mv, mode

    def keep_coef(self, on=True):
        self._chk(self.lib.icsp_debug_keep_coef(self.ctx, int(on)), "icsp_debug_keep_coef")

    def upload_syntax(self, first, levels, mpm, mvd):
        lv = np.ascontiguousarray(levels, np.int16); mp = np.ascontiguousarray(mpm, np.uint8); mv = np.ascontiguousarray(mvd, np.int8)
        self._chk(self.lib.icsp_upload_syntax(self.ctx, first, lv.shape[0], _vp(lv), _vp(mp), _vp(mv)), "icsp_upload_syntax")

    def decode_resident(self, first, n):
        """Decoder reconstruction (DEC:2083-2272) of the resident syntax of slots [first, first+n) into their recon planes."""
        self._chk(self.lib.icsp_decode_resident(self.ctx, first, n), "icsp_decode_resident")

    def pack_bits(self, first, n, out=None):
        """Device bit packer on the encoded slots [first, first+n): (body bytes as uint8 array, bit count).
        `out`: optional reusable uint8 host buffer (icsp_bitstream_bound bytes always suffice)."""
        if out is None:
            out = np.empty(self.lib.icsp_bitstream_bound(C.byref(self.params), n), np.uint8)
        nbits = C.c_uint64(0)
        self._chk(self.lib.icsp_pack_bits(self.ctx, first, n, _vp(out), out.size, C.byref(nbits)), "icsp_pack_bits")
        return out[: (nbits.value + 7) // 8], nbits.value

    def pack_count(self, first, n) -> int:
        """Lengths + prefix sums of slots [first, first+n) on the device: the bit count of their string."""
        nbits = C.c_uint64(0)
        self._chk(self.lib.icsp_pack_count(self.ctx, first, n, C.byref(nbits)), "icsp_pack_count")
        return nbits.value

    def pack_into(self, first, n, at_bit, body_image: np.ndarray):
        """Packs the range icsp_pack_count last measured at bit `at_bit` of the (zero-initialised) body image."""
        self._chk(self.lib.icsp_pack_into(self.ctx, first, n, at_bit, _vp(body_image), body_image.size), "icsp_pack_into")

    def host_warm(self, pinned: np.ndarray):
        """Zero bytes (at most 16 MB) written into a pinned range by DMA (icsp_host_warm)."""
        self._chk(self.lib.icsp_host_warm(self.ctx, _vp(pinned), pinned.size), "icsp_host_warm")

    def prepare(self):
        self._chk(self.lib.icsp_prepare(self.ctx), "icsp_prepare")

    def upload_sync(self, frames: np.ndarray, first=0):
        """Upload on the device's shared upload stream, back when the frames are there (icsp_upload_sync)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        self._chk(self.lib.icsp_upload_sync(self.ctx, _vp(frames), first, frames.shape[0]), "icsp_upload_sync")

    def set_groups(self, p_groups=0, i_groups=0):
        """GOP groups on separate streams / parts of a large all-intra batch for this context (icsp_set_groups; 0 keeps)."""
        self._chk(self.lib.icsp_set_groups(self.ctx, p_groups, i_groups), "icsp_set_groups")

    def last_choice(self) -> dict:
        """What the last encode_resident chose (icsp_debug_last_choice)."""
        v = [C.c_int(0) for _ in range(6)]
        self._chk(self.lib.icsp_debug_last_choice(self.ctx, *[C.byref(x) for x in v]), "icsp_debug_last_choice")
        return {"intra_lanes_per_block": v[0].value, "intra_waves_per_workgroup": v[1].value, "intra_recon_ring": bool(v[2].value),
                "range_whole_on_one_stream": bool(v[3].value), "gop_groups": v[4].value, "intra_rows_chained": v[5].value}

    def single_stream(self, on=True):
        """Every kernel of the context on its one stream (icsp_single_stream)."""
        self._chk(self.lib.icsp_single_stream(self.ctx, int(on)), "icsp_single_stream")

    def copy_streams(self, shared=True):
        """Uploads and downloads on the device's two shared transfer streams (icsp_copy_streams)."""
        self._chk(self.lib.icsp_copy_streams(self.ctx, 1 if shared else 0), "icsp_copy_streams")

    def pack_bitstream(self, first, n) -> bytes:
        """The .bin image of slots [first, first+n) with the body packed on the device."""
        p = self.params
        return assemble_bitstream(p.width, p.height, p.qp_dc, p.qp_ac, p.intra_period, [self.pack_bits(first, n)])

    def download_coef(self, first, n):
        c = np.zeros((n, self.nmb, 6, 64), np.float64)
        self._chk(self.lib.icsp_download_coef(self.ctx, first, n, _vp(c)), "icsp_download_coef")
        return c

    def device_view(self) -> DeviceView:
        v = DeviceView()
        self._chk(self.lib.icsp_device_view(self.ctx, C.byref(v)), "icsp_device_view")
        return v

    def profile(self, on=True, only=None):
        """on=True: HIP events around every kernel; only=[names]: just those kernels (cheaper inside a timed region)."""
        flag = int(bool(on))
        if on and only:
            flag = 0
            for name in only:
                flag |= 1 << (KERNELS.index(name) + 1)
        self._chk(self.lib.icsp_profile_enable(self.ctx, flag), "icsp_profile_enable")
        self._chk(self.lib.icsp_profile_reset(self.ctx), "icsp_profile_reset")

    def profile_get(self) -> dict:
        out = {}
        for i, name in enumerate(KERNELS):
            ms = C.c_double(0); n = C.c_longlong(0)
            self._chk(self.lib.icsp_profile_get(self.ctx, i, C.byref(ms), C.byref(n)), "icsp_profile_get")
            out[name] = (ms.value, n.value)
        return out
