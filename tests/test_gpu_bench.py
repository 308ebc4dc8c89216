"""bench.py contract on the GPU box: one JSON line with the required keys at N=1, and the N>1 code path (barrier,
max-over-ranks, aggregate value) exercised with two ranks sharing the only GPU (gloo for the tiny control traffic)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "frames/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 1e4 and abs(d["value"] - 300 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 0.01
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["psnr_y_db"] > 25 and d["ippp"]["value"] > 1e4
    assert d["ranks_seen"] == 1 and all(d["parity"].values()) and len(d["parity"]) == 5
    # the 8-GPU workloads of BASELINE configs[3] and [4], here on one GPU, outputs checked against the reference / the oracle
    assert d["config4"]["frames"] == 3390 and d["config4"]["recon_equals_reference"] and d["config4"]["value"] > 1e4
    assert d["config4"]["all_intra_loaded"]["value"] > 1e4
    assert d["config5"]["frames"] == 3000 and d["config5"]["recon_equals_oracle"] and d["config5"]["value"] > 100
    assert d["cpu_baseline"].get("single_thread") is None or d["cpu_baseline"]["single_thread"]["value"] > 0
    assert d["e2e"]["rc"] == 0 and d["e2e"]["bin_equals_reference"] and d["e2e"]["recon_equals_reference"]


def test_two_ranks_launched_the_drivers_way():
    """torch.distributed.run starts the ranks (how the driver runs N > 1); both share the only GPU here."""
    env = dict(os.environ, ICSP_BENCH_BACKEND="gloo", ICSP_BENCH_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--legs", "ippp"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    # value is the aggregate over both ranks: 2 x 300 frames per step
    assert abs(d["value"] - 2 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks and the
    strong-scaling legs shard their GOPs over them (339 GOPs -> 170 + 169; 100 GOPs -> 50 + 50)."""
    env = dict(os.environ, ICSP_BENCH_BACKEND="gloo", ICSP_BENCH_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--legs", "config4,config5"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["config4"]["scaling"] == "strong" and d["config4"]["frames"] == 3390 and d["config4"]["recon_equals_reference"]
    assert d["config5"]["scaling"] == "strong" and d["config5"]["frames"] == 3000 and d["config5"]["recon_equals_oracle"]


def test_eight_ranks_on_one_gpu_the_drivers_scaling_shape():
    """The shape of the driver's 8-GPU run -- eight ranks, the strong-scaling legs sharded 8 ways (339 GOPs -> 43/42 per rank,
    100 GOPs of 1088p -> 13/12 per rank) -- with all eight ranks on the one GPU of this box (gloo for the control traffic): memory
    of eight contexts per leg, wall time well inside the driver's limit, every rank seen, every rank's regime in the line.  Not a
    scaling measurement (the ranks share a device); no scaling curve over GPUs has been measured on hardware."""
    import time
    env = dict(os.environ, ICSP_BENCH_BACKEND="gloo", ICSP_BENCH_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--repeats", "2",
                        "--legs", "ippp,config4,config5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1700)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    d = _last_json(r.stdout)
    assert wall < 1500, wall
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 8 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    assert d["ippp"]["value"] > 1e4
    c4, c5 = d["config4"], d["config5"]
    assert c4["scaling"] == "strong" and c4["frames"] == 3390 and c4["recon_equals_reference"] and c4["regime"]["gops_per_rank"] in (42, 43)
    assert c5["scaling"] == "strong" and c5["frames"] == 3000 and c5["recon_equals_oracle"] and c5["regime"]["gops_per_rank"] in (12, 13)
    for leg in (d, d["ippp"], c4, c5, c4["all_intra_loaded"]):
        assert "intra_lanes_per_block" in leg["regime"] and "gop_groups" in leg["regime"]
    assert all(d["parity"].values())
