"""N>1 host path on CPU: two gloo ranks shard closed GOPs, rank 0 gathers in frame order and packs the bitstream."""
import json
import os
import subprocess
import sys

import pytest

from icspcodec_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gop_shards_cover_all_frames_on_gop_boundaries():
    for n, p, w in [(300, 10, 8), (300, 0, 8), (12, 6, 2), (13, 6, 2), (5, 3, 4), (90, 10, 8), (7, 0, 3), (1, 10, 2)]:
        sh = shard.gop_shards(n, p, w)
        assert len(sh) == w
        pos = 0
        for first, cnt in sh:
            if cnt:
                assert first == pos and first % max(p, 1) == 0
                pos += cnt
        assert pos == n
        L = max(p, 1)
        gops = [(c + L - 1) // L for _, c in sh]
        assert max(gops) - min(gops) <= 1          # whole GOPs, balanced to within one


@pytest.mark.parametrize("name,n,qp,period", [("foremanlike", 12, 16, 6), ("foremanlike", 12, 16, 0), ("stefanlike", 3, 8, 3)])
def test_two_rank_gloo_equals_reference_stream(tmp_path, golden_dir, name, n, qp, period):
    out = tmp_path / "r.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 400)), os.path.join(ROOT, "tests", "_dist_worker.py"),
           name, str(n), str(qp), str(period), str(out)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    got = json.load(open(out))
    assert got["world"] == 2
    streams = json.load(open(os.path.join(golden_dir, "streams.json")))
    ref = [s for s in streams if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == (name, n, qp, period) and "bin_sha256" in s][0]
    assert got["bin_sha256"] == ref["bin_sha256"] and got["bin_bytes"] == ref["bin_bytes"]
    assert got["recon_sha256"] == ref["recon_sha256"]
