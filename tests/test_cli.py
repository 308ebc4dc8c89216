"""The C++ host program keeps the reference's command line (encoder_main.cpp, ICSP_Codec_Encoder_source.cpp:84-176)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from icspcodec_amd import clipgen
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC = os.path.join(ROOT, "icspcodec_amd", "icsp_enc")


def run(args, cwd):
    return subprocess.run([ENC] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)


def test_option_errors_match_reference_messages_and_exit_code(tmp_path):
    cases = [([], b"[ERROR] unenough parameters in parsing_command\n"),
             (["-i", "a_b.yuv", "-w", "352"], b"[ERROR] uncorrect parameters in parsing_command\n"),
             (["--bogus", "1"], b"[ERROR] uncorrect parameters in parsing_command\n"),
             (["-i", "missing_x.yuv", "-n", "1", "-q", "16"], b"fail to load cif.yuv\n error from YCbCrLoad\n")]
    for args, msg in cases:
        r = run(args, tmp_path)
        assert r.returncode == 255 and r.stdout == msg, (args, r.stdout)
        if os.path.exists(po.REF_ENC):          # where the real reference is built, it says the same
            rr = subprocess.run([po.REF_ENC] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            assert rr.returncode == 255 and rr.stdout == msg, (args, rr.stdout)
    r = run(["-h"], tmp_path)
    assert r.returncode == 0 and r.stdout.startswith(b"usage: ./ICSPCodec [option] [values]\n")
    r = run(["--help"], tmp_path)
    assert r.returncode == 0 and b"--intraPeriod: period of intra frame(0: All intra)" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("period,extra", [(6, []), (0, []), (1, []), (6, ["--EnMultiThread", "1"]), (6, ["--hostpack"]),
                                          (0, ["--hostpack"]), (6, ["--staged"]), (0, ["--staged", "--chunk", "5"]),
                                          (6, ["--chunk", "6", "--streams", "2"]), (6, ["--staged", "--hostpack", "--chunk", "1"]),
                                          (6, ["--chunk", "6", "--streams", "2", "--binest", "200000"]),
                                          (0, ["--chunk", "3", "--streams", "3", "--binest", "4096"]),
                                          (6, ["--nopin", "--chunk", "6", "--streams", "2"]), (0, ["--nopin"])])
def test_cli_outputs_equal_reference_files(tmp_path, golden_dir, period, extra):
    n, qp = 12, 16
    clip = clipgen.synth_clip("foremanlike", n)
    fn = clipgen.file_name("foremanlike", n)
    clip.tofile(tmp_path / fn)
    r = run(["-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period)] + extra, tmp_path)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.decode().splitlines()
    assert lines == ["Encoding FRAME_%03d(%c) done!" % (f, "I" if period == 0 or f % period == 0 else "P") for f in range(n)]
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    ref = [s for s in streams if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", n, qp, period) and "bin_sha256" in s][0]
    binp = tmp_path / f"foremanlike_compCIF_{qp}_{qp}_{period}.bin"
    assert hashlib.sha256(binp.read_bytes()).hexdigest() == ref["bin_sha256"]
    assert hashlib.sha256((tmp_path / "test_yuv.yuv").read_bytes()).hexdigest() == ref["recon_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,qp,period,extra", [("foremanlike", 16, 0, ["--chunk", "64"]), ("foremanlike", 16, 10, ["--chunk", "50"]),
                                                  ("stefanlike", 8, 10, ["--chunk", "40", "--streams", "3"]),
                                                  ("foremanlike", 16, 10, ["--chunk", "50", "--staged"]),
                                                  ("foremanlike", 16, 0, ["--chunk", "64", "--streams", "1"])])
def test_cli_300_frames_in_ramped_chunks(tmp_path, golden_dir, name, qp, period, extra):
    """300 frames in many chunks: the first chunk a quarter and the second a half of the rest, two (or three) workers whose
    uploads go through the device's uploader thread while they pack and download (one worker: on its own stream) ->
    the reference's files."""
    n = 300
    clip = clipgen.synth_clip(name, n)
    fn = clipgen.file_name(name, n)
    clip.tofile(tmp_path / fn)
    r = run(["-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period), "--stats"] + extra, tmp_path)
    assert r.returncode == 0, r.stdout
    st = json.loads(r.stdout.decode().split("[icsp_enc]", 1)[1])
    assert st["chunks"] >= 6 and st["workers"] == (int(extra[extra.index("--streams") + 1]) if "--streams" in extra else 2)
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    ref = [s for s in streams if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == (name, n, qp, period) and "bin_sha256" in s][0]
    assert hashlib.sha256((tmp_path / f"{name}_compCIF_{qp}_{qp}_{period}.bin").read_bytes()).hexdigest() == ref["bin_sha256"]
    assert hashlib.sha256((tmp_path / "test_yuv.yuv").read_bytes()).hexdigest() == ref["recon_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n,period,extra", [(704, 576, 6, 3, []), (64, 48, 40, 4, ["--chunk", "8", "--streams", "2"]),
                                                (1920, 1088, 4, 2, ["--chunk", "2"]), (32, 16, 9, 0, ["--staged"])])
def test_cli_other_frame_sizes(tmp_path, w, h, n, period, extra):
    """--width / --height (the reference hard-codes 352x288, encoder_main.cpp:20): the files equal what the oracle and the host
    bit writer give for that geometry."""
    from oracle import pyoracle as po
    from icspcodec_amd import capi
    qp = 16
    clip = clipgen.synth_clip("mobilelike", n, width=w, height=h)
    fn = f"mobilelike_x({w}X{h})_{n}f.yuv"
    clip.tofile(tmp_path / fn)
    r = run(["-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period), "--width", str(w), "--height", str(h)] + extra, tmp_path)
    assert r.returncode == 0, r.stdout
    want = po.encode_sequence(clip, w, h, qp, qp, period, nthreads=4)
    bs = capi.write_bitstream(w, h, qp, qp, period, want["levels"], want["acflag"], want["mpm"], want["mvd"])
    assert (tmp_path / f"mobilelike_compCIF_{qp}_{qp}_{period}.bin").read_bytes() == bs
    assert np.array_equal(np.fromfile(tmp_path / "test_yuv.yuv", np.uint8), want["recon"].reshape(-1))


DEC = os.path.join(ROOT, "icspcodec_amd", "icsp_dec")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--gpus", "2"], ["--gpus", "3", "--hostpack"], ["--EnMultiThread", "2"], ["--EnMultiThread", "4"],
                                   ["--EnMultiThread", "2", "--staged"]])
def test_cli_multi_shard_path_on_one_gpu(tmp_path, golden_dir, extra):
    """--gpus N / --EnMultiThread N with more shards than devices (shards share devices round-robin): closed-GOP shards in
    their own contexts and threads, one bit-string piece per shard concatenated bit-wise on the host -> the same files as
    the reference's single run."""
    n, qp, period = 12, 16, 6                      # 2 GOPs: --gpus 3 is clipped to 2 shards
    clip = clipgen.synth_clip("foremanlike", n)
    fn = clipgen.file_name("foremanlike", n)
    clip.tofile(tmp_path / fn)
    r = subprocess.run([ENC, "-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period)] + extra, cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    ref = [s for s in streams if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", n, qp, period) and "bin_sha256" in s][0]
    assert hashlib.sha256((tmp_path / f"foremanlike_compCIF_{qp}_{qp}_{period}.bin").read_bytes()).hexdigest() == ref["bin_sha256"]
    assert hashlib.sha256((tmp_path / "test_yuv.yuv").read_bytes()).hexdigest() == ref["recon_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 1])
def test_decoder_cli_writes_reference_files(tmp_path, golden_dir, idx):
    """icsp_enc -> icsp_dec with the reference decoder's command line (decode.cpp:4-27): the decoded file and the PSNR line
    equal what the reference decoder binary produced for the same stream (tests/golden/decoded.json)."""
    import re
    case = json.load(open(os.path.join(golden_dir, "decoded.json")))[idx]
    name, n, qp, period = case["clip"], case["nframes"], case["qp"], case["intra_period"]
    clip = clipgen.synth_clip(name, n)
    fn = clipgen.file_name(name, n)
    os.makedirs(tmp_path / "output"); os.makedirs(tmp_path / "data")
    clip.tofile(tmp_path / "data" / fn)
    r = subprocess.run([ENC, "-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period)], cwd=tmp_path / "data",
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout
    binname = f"{name}_compCIF_{qp}_{qp}_{period}.bin"
    bs = (tmp_path / "data" / binname).read_bytes()
    assert hashlib.sha256(bs).hexdigest() == case["bin_sha256"]
    (tmp_path / "output" / binname).write_bytes(bs)
    r = subprocess.run([DEC, str(n), binname, str(qp), str(qp), str(period), fn], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout
    out = tmp_path / ("check_test_intra_yuv.yuv" if period == 1 else "check_test_inter_yuv.yuv")
    assert hashlib.sha256(out.read_bytes()).hexdigest() == case["decoded_sha256"]
    line = (tmp_path / "experimental_Result_Decoding.txt").read_text()
    m = re.fullmatch(r"decoding time: [0-9.]+\(s\) PSNR: ([0-9.]+) QPDC: (\d+)  QPAC: (\d+) Period: (\d+)\n", line)
    assert m and abs(float(m.group(1)) - case["psnr"]) < 1.5e-4 and (int(m.group(2)), int(m.group(3)), int(m.group(4))) == (qp, qp, period)
    r = subprocess.run([DEC, str(n), "nope.bin", "1", "1", "1", fn], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 255 and r.stdout.startswith(b"fail to load output")


@pytest.mark.gpu
@pytest.mark.parametrize("period,extra", [(10, ["--gpus", "2", "--chunk", "30", "--streams", "2"]), (0, ["--gpus", "2", "--chunk", "25", "--streams", "2"]),
                                          (10, ["--gpus", "3", "--chunk", "20"]), (10, ["--gpus", "2", "--chunk", "30", "--staged"])])
def test_cli_two_device_records_on_one_gpu(tmp_path, golden_dir, period, extra):
    """ICSP_FAKE_DEVICES=N: the library presents N devices (all the one GPU of this box) with per-device records of their own, so
    icsp_enc --gpus N runs its per-device machinery -- chunk c on device c mod N, one uploader thread and one pair of shared
    transfer streams per device, the search tables uploaded per device -- with several device records: the same files as the
    reference's single run, and the statistics say how many devices were used.  (The closest a one-GPU box gets to --gpus 2;
    nothing here has run on two physical devices.)"""
    n, qp = 300, 16 if period == 0 else 8
    name = "foremanlike" if period == 0 else "stefanlike"
    clip = clipgen.synth_clip(name, n)
    fn = clipgen.file_name(name, n)
    clip.tofile(tmp_path / fn)
    ndev = int(extra[1])
    env = dict(os.environ, ICSP_FAKE_DEVICES=str(ndev))
    r = subprocess.run([ENC, "-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period), "--stats"] + extra, cwd=tmp_path, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:]
    st = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")][0][len("[icsp_enc]"):])
    assert st["devices"] == ndev and st["workers"] >= ndev and st["chunks"] >= 2 * ndev
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    ref = [s for s in streams if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == (name, n, qp, period) and "bin_sha256" in s][0]
    assert hashlib.sha256((tmp_path / f"{name}_compCIF_{qp}_{qp}_{period}.bin").read_bytes()).hexdigest() == ref["bin_sha256"]
    assert hashlib.sha256((tmp_path / "test_yuv.yuv").read_bytes()).hexdigest() == ref["recon_sha256"]
