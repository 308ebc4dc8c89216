"""Pinned host ranges the caller owns (icsp_host_register / icsp_host_unregister / icsp_host_warm): uploads from and downloads
into a registered mapping of a file give the same bytes as plain arrays; the warm-up writes zeros only."""
import mmap
import os

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen

pytestmark = pytest.mark.gpu
W, H = 352, 288


def test_registered_file_mappings_carry_the_same_bytes(tmp_path):
    n, q, period = 12, 16, 6
    clip = clipgen.synth_clip("foremanlike", n)
    ref = capi.Encoder(W, H, q, q, period, max_frames=n)
    want = ref.encode(clip)
    ref.close()
    fin, fout = tmp_path / "in.yuv", tmp_path / "out.yuv"
    clip.tofile(fin)
    with open(fout, "wb") as f:
        f.truncate(clip.nbytes)
    with open(fin, "rb") as fi, open(fout, "r+b") as fo:
        mi = mmap.mmap(fi.fileno(), 0, access=mmap.ACCESS_READ)
        mo = mmap.mmap(fo.fileno(), 0, access=mmap.ACCESS_WRITE)
        src = np.frombuffer(mi, np.uint8)
        dst = np.frombuffer(mo, np.uint8)
        pinned_in = capi.host_register(src, read_only=True)
        pinned_out = capi.host_register(dst)
        enc = capi.Encoder(W, H, q, q, period, max_frames=n)
        enc.prepare()
        rc = enc.lib.icsp_upload(enc.ctx, capi._vp(src), 0, n)
        assert rc == 0
        enc.encode_resident(0, n)
        rc = enc.lib.icsp_download(enc.ctx, 0, n, None, None, None, None, capi._vp(dst))
        assert rc == 0
        got = enc.download(0, n, what=("levels", "mvd"))
        enc.close()
        assert np.array_equal(got["levels"], want["levels"]) and np.array_equal(got["mvd"], want["mvd"])
        assert np.array_equal(dst.reshape(want["recon"].shape), want["recon"])
        if pinned_in:
            assert capi.host_unregister(src)
        if pinned_out:
            assert capi.host_unregister(dst)
        del src, dst
        mi.close(); mo.close()
    assert np.array_equal(np.fromfile(fout, np.uint8).reshape(want["recon"].shape), want["recon"])      # the file itself


def test_host_warm_writes_zeros_only_and_keeps_a_count_from_leaking():
    enc = capi.Encoder(W, H, 16, 16, 0, max_frames=4)
    clip = clipgen.synth_clip("foremanlike", 4)
    enc.upload(clip)
    enc.encode_resident(0, 4)
    bits = enc.pack_count(0, 4)
    assert bits > 0
    raw = np.empty((3 << 20) + 8192, np.uint8)            # whole pages of its own (icsp_hip.h, icsp_host_register)
    buf = raw[(-raw.ctypes.data) % 4096:][: 3 << 20]
    buf[:] = 0xAB
    assert capi.host_register(buf)
    enc.host_warm(buf[4096:])
    assert not buf[4096:].any() and (buf[:4096] == 0xAB).all()
    with pytest.raises(RuntimeError):                    # the scratch no longer holds the counted string
        enc.pack_into(0, 4, 0, np.zeros(1 << 20, np.uint8))
    assert enc.pack_count(0, 4) == bits
    image = np.zeros(bits // 8 + 64, np.uint8)
    enc.pack_into(0, 4, 0, image)
    want, wbits = enc.pack_bits(0, 4)
    assert wbits == bits and np.array_equal(image[: len(want)], want)
    assert capi.host_unregister(buf)
    enc.close()


def test_shared_transfer_streams_from_two_threads():
    """icsp_copy_streams: two contexts on one device, a host thread each, uploads on the device's shared upload stream and
    downloads (reconstruction, bit strings, syntax arrays) on its shared download stream, chunk after chunk -- the same bytes as
    one context on its own stream; switching back to the own stream works too."""
    import threading
    n, q, period, cn = 48, 8, 6, 12
    clip = clipgen.synth_clip("stefanlike", n)
    ref = capi.Encoder(W, H, q, q, period, max_frames=n)
    want = ref.encode(clip)
    want_bin = ref.pack_bitstream(0, n)
    ref.close()
    got = {k: np.zeros_like(v) for k, v in want.items()}
    bits, strings = {}, {}
    errs = []

    def work(widx):
        try:
            enc = capi.Encoder(W, H, q, q, period, max_frames=cn)
            enc.prepare()
            enc.copy_streams(True)
            for c in range(widx, n // cn, 2):
                a = c * cn
                enc.upload(clip[a:a + cn], 0)
                enc.encode_resident(0, cn)
                b = enc.pack_count(0, cn)
                img = np.zeros((b + 7) // 8 + 64, np.uint8)
                enc.pack_into(0, cn, 3, img)                       # at bit 3: head and tail bytes through the scratch
                o = enc.download(0, cn)
                for k in got:
                    got[k][a:a + cn] = o[k]
                bits[c], strings[c] = b, img
                if c == 1:
                    enc.copy_streams(False)                        # the rest of this worker's chunks on its own stream
            enc.close()
        except Exception as e:           # noqa: BLE001
            errs.append((widx, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    # the chunks' strings, each packed at bit 3 of its own image, put together at their offsets == the whole body
    body = np.zeros(len(want_bin) + 64, np.uint8)
    at = 0
    for c in range(n // cn):
        s = np.unpackbits(strings[c])[3:3 + bits[c]]
        tgt = np.unpackbits(body)
        tgt[at:at + bits[c]] |= s
        body = np.packbits(tgt)
        at += bits[c]
    assert capi.finish_image(W, H, q, q, period, np.concatenate([np.zeros(14, np.uint8), body]), at) == want_bin


def test_upload_sync_from_a_second_thread_beside_the_downloads():
    """icsp_upload_sync: the next chunk's frames go up from another host thread while the context's own thread packs and
    downloads the current chunk (icsp_enc's uploader thread); refused without the shared transfer streams."""
    import threading
    n, q, period, cn = 36, 16, 3, 9
    clip = clipgen.synth_clip("mobilelike", n)
    ref = capi.Encoder(W, H, q, q, period, max_frames=n)
    want = ref.encode(clip)
    ref.close()
    enc = capi.Encoder(W, H, q, q, period, max_frames=cn)
    with pytest.raises(RuntimeError):
        enc.upload_sync(clip[:cn])
    enc.copy_streams(True)
    got = {k: np.zeros_like(v) for k, v in want.items()}
    enc.upload_sync(clip[:cn])
    for c in range(n // cn):
        a = c * cn
        enc.encode_resident(0, cn)
        enc.pack_count(0, cn)                                  # the kernels are through: the frames may be overwritten
        up = None
        if c + 1 < n // cn:
            up = threading.Thread(target=enc.upload_sync, args=(clip[a + cn:a + 2 * cn],))
            up.start()
        o = enc.download(0, cn)
        for k in got:
            got[k][a:a + cn] = o[k]
        if up:
            up.join()
    enc.close()
    for k in want:
        assert np.array_equal(got[k], want[k]), k


# ---- icsp_encode_gop as a pipeline (chunks of whole GOPs: upload, kernels and download side by side): pageable, pinned and
#      registered caller memory give the bytes of upload + encode_resident + download
def _plain(clip, q, period):
    n = clip.shape[0]
    ref = capi.Encoder(W, H, q, q, period, max_frames=n)
    ref.upload(clip)
    ref.encode_resident(0, n)
    want = ref.download(0, n)
    body, bits = ref.pack_bits(0, n)
    body = body.copy()
    ref.close()
    return want, body, bits


def _same(got, want, tag):
    for k in ("levels", "acflag", "mpm", "mvd", "recon"):
        assert np.array_equal(np.asarray(got[k]).reshape(want[k].shape), want[k]), tag + k


@pytest.mark.parametrize("period,n", [(0, 230), (10, 205), (7, 150)])
@pytest.mark.parametrize("mem", ["pageable", "pinned", "registered"])
def test_encode_gop_pipeline_matches_the_resident_path(mem, period, n):
    q = 16 if period == 0 else 8
    clip = clipgen.synth_clip("stefanlike" if period else "foremanlike", n)
    want, body, bits = _plain(clip, q, period)
    nmb, fsz = (W // 16) * (H // 16), W * H * 3 // 2
    shapes = dict(levels=((n, nmb, 6, 64), np.int16), acflag=((n, nmb, 6), np.uint8), mpm=((n, nmb, 4), np.uint8),
                  mvd=((n, nmb, 2), np.int8), recon=((n, fsz), np.uint8))
    enc = capi.Encoder(W, H, q, q, period, max_frames=n)
    if mem == "pinned":
        src = capi.host_alloc_array(clip.shape, np.uint8)
        src[:] = clip
        out = {k: capi.host_alloc_array(s, d) for k, (s, d) in shapes.items()}
    elif mem == "registered":
        # whole pages of their own: a registration covers whole pages, and the runtime treats any buffer that STARTS inside a registered
        # page as pinned -- a heap array next to an unaligned registered one was DMA'd into as if pinned and the GPU faulted on its first
        # page beyond the registration ("Memory access fault ... on address 0x56dd7076e000", intermittently, round 4)
        keep = []

        def aligned(shape, dtype):
            nb = int(np.prod(shape)) * np.dtype(dtype).itemsize
            raw = np.empty(((nb + 4095) // 4096 + 2) * 4096, np.uint8)
            off = (-raw.ctypes.data) % 4096
            keep.append((raw, raw[off: off + ((nb + 4095) // 4096) * 4096]))
            return raw[off: off + nb].view(dtype).reshape(shape)
        src = aligned(clip.shape, np.uint8)
        src[:] = clip
        out = {k: aligned(s, d) for k, (s, d) in shapes.items()}
        assert capi.host_register(keep[0][1], read_only=True)
        for _, pages in keep[1:]:
            assert capi.host_register(pages)
    else:
        src = clip.copy()
        out = {k: np.full(s, 0x55, d) for k, (s, d) in shapes.items()}
    for a in out.values():
        a.reshape(-1).view(np.uint8)[:] = 0x55
    try:
        for rep in range(2):                                  # the second call finds the staging buffers and streams in place
            got = enc.encode(src, out=out)
            _same(got, want, f"{mem} p={period} n={n} rep={rep}: ")
        # outputs the caller does not want
        part = dict(out)
        for a in out.values():
            a.reshape(-1).view(np.uint8)[:] = 0
        rc = enc.lib.icsp_encode_gop(enc.ctx, capi._vp(src), n, None, None, capi._vp(out["mpm"]), None, capi._vp(out["recon"]))
        assert rc == 0
        assert np.array_equal(out["recon"], want["recon"]) and np.array_equal(out["mpm"], want["mpm"]) and not out["levels"].any()
        # the packed form: body bits + reconstruction only
        out["recon"][:] = 0
        b2, nb2 = enc.encode_packed(src, recon=out["recon"])
        assert nb2 == bits and np.array_equal(b2, body[: len(b2)]) and np.array_equal(out["recon"], want["recon"])
        b3, nb3 = enc.encode_packed(src)
        assert nb3 == bits and np.array_equal(b3, body[: len(b3)])
        # and the resident calls still work on a context whose transfers now run on the shared streams
        enc.upload(clip[:20])
        enc.encode_resident(0, 20)
        assert np.array_equal(enc.download(0, 20, what=("recon",))["recon"], want["recon"][:20])
    finally:
        enc.close()
        if mem == "registered":
            for _, pages in keep:
                assert capi.host_unregister(pages)
        if mem == "pinned":
            capi.host_free_array(src)
            for a in out.values():
                capi.host_free_array(a)


def test_encode_gop_small_batches_take_the_single_chunk_path():
    for period, n in ((0, 5), (4, 9), (10, 3)):
        clip = clipgen.synth_clip("mobilelike", n)
        want, body, bits = _plain(clip, 8, period)
        enc = capi.Encoder(W, H, 8, 8, period, max_frames=n)
        _same(enc.encode(clip), want, f"p={period} n={n}: ")
        b, nb = enc.encode_packed(clip)
        enc.close()
        assert nb == bits and np.array_equal(b, body[: len(b)])


def test_encode_gop_large_frames_pipeline():
    """1088p: a chunk is one GOP; three GOPs, ragged last one"""
    w, h, period, n = 1920, 1088, 4, 10
    clip = clipgen.synth_clip("tablelike", n, width=w, height=h)
    ref = capi.Encoder(w, h, 16, 16, period, max_frames=n)
    ref.upload(clip)
    ref.encode_resident(0, n)
    want = ref.download(0, n)
    ref.close()
    enc = capi.Encoder(w, h, 16, 16, period, max_frames=n)
    got = enc.encode(clip)
    enc.close()
    for k in ("levels", "acflag", "mpm", "mvd", "recon"):
        assert np.array_equal(got[k], want[k]), k


# ---- caller memory the library does not KNOW to be pinned never reaches the runtime as a plain pointer (the
#      round-4 device fault).  The session runs WITHOUT the mallopt hook round 4's conftest.py had.
def test_register_refuses_a_range_that_does_not_start_on_a_page():
    raw = np.zeros(3 * 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    lib = capi.load()
    assert lib.icsp_host_register(capi._vp(raw[off + 8:]), 4096, 0) == 2          # ICSP_ERR_UNCORRECT_PARAM
    assert lib.icsp_host_register(capi._vp(raw[off:]), 5000, 0) == 0              # a ragged length is rounded up to the page
    assert lib.icsp_host_unregister(capi._vp(raw[off:])) == 0


def test_plain_buffer_that_starts_inside_a_registered_page_is_staged():
    """Round 4's trigger: the runtime reports any pointer inside a registered page as pinned, so a plain buffer that starts there and
    ends beyond the registration was DMA'd as if pinned and the device faulted where the registration ends.  Both directions."""
    n, q = 64, 16
    clip = clipgen.synth_clip("foremanlike", n)
    ref = capi.Encoder(W, H, q, q, 0, max_frames=n)
    want = ref.encode(clip)
    ref.close()
    fsz = clip.shape[1]
    nb = n * fsz
    half = 16 * 4096
    span = ((half + nb + 4095) // 4096 + 1) * 4096                              # a registered head of `half` bytes + a plain buffer that starts inside it
    raw = np.empty(2 * span + 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    pa, pb = raw[off: off + span], raw[off + span: off + 2 * span]              # one such arrangement for the source, one for the target
    assert capi.host_register(pa[:half]) and capi.host_register(pb[:half])
    try:
        src = pa[half - 100: half - 100 + nb].reshape(n, fsz)                      # starts 100 bytes before the registration ends
        src[:] = clip
        enc = capi.Encoder(W, H, q, q, 0, max_frames=n)
        assert enc.lib.icsp_upload(enc.ctx, capi._vp(src), 0, n) == 0
        enc.encode_resident(0, n)
        dst = pb[half - 4000: half - 4000 + nb].reshape(n, fsz)
        dst[:] = 0
        assert enc.lib.icsp_download(enc.ctx, 0, n, None, None, None, None, capi._vp(dst)) == 0
        assert np.array_equal(dst, want["recon"])
        got = enc.encode(src)                                                      # the one-call pipeline decides per array too
        enc.close()
        _same(got, want, "straddling source: ")
    finally:
        assert capi.host_unregister(pa[:half]) and capi.host_unregister(pb[:half])


def test_host_warm_refuses_plain_memory():
    enc = capi.Encoder(W, H, 16, 16, 0, max_frames=2)
    plain = np.zeros(1 << 20, np.uint8)
    assert enc.lib.icsp_host_warm(enc.ctx, capi._vp(plain), plain.size) == 2       # ICSP_ERR_UNCORRECT_PARAM
    enc.close()


def test_fault_reproducer_runs_clean_bounded():
    """tools/repro_fault.py (heap churn of 50-300 MB frame arrays, plain pointers, contexts created and destroyed, registered ranges in
    play) for a few seconds per mode, each in a process of its own: no device fault, every checked transfer exact."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("plain", "regsplit"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "repro_fault.py"), mode, "4"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (mode, r.returncode, r.stderr[-400:])
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["status"] == "ok" and line["rounds"] >= 2, line
