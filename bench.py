#!/usr/bin/env python3
"""bench.py — CIF encode throughput of the HIP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts N ranks itself (a child `python -m torch.distributed.run
... bench.py --gpus N ...`, before anything here touches the GPU) and exits with the child's code; started BY
torch.distributed.run (the driver's way) it is one rank.  One rank per GPU, RCCL only for the barrier and the reductions of
the timing: closed GOPs / independent frames shard with no data-path collective.

A "step" is one pass of the per-macroblock encode loop over one resident batch.  `value` is BASELINE.json configs[1]: a
300-frame CIF clip, all-intra, QP 16 (synthetic `foremanlike`; the bundled clips are absent from the reference checkout),
every rank its own frames (weak scaling), inputs uploaded to HBM before the timed region.  The timed steps ALTERNATE between
two disjoint resident 300-frame batches of different content (A = frames 0-299 of the sequence, B = frames 300-599; A, B, A,
B ...): what a host streaming a clip chunk by chunk issues -- independent ranges, which the library does not order against
each other -- and not the re-encode of one range.  `isolated_pass` is one pass with the host waiting before and after.
Rank 0 prints ONE JSON line.  In the same line:
  ippp       configs[2]: `stefanlike` 300 f, --intraPeriod 10, QP 8 (motion search + compensation), weak, like `value`
  config4    configs[3]: the twelve CIF clips (3390 frames, 339 closed GOPs), --intraPeriod 10, QP 16, ONE batch whose GOPs
             are sharded over the ranks (strong scaling); `all_intra_loaded`: the same 3390 frames all-intra (a loaded chip)
  config5    configs[4]: 1920x1088, --intraPeriod 30, 3000 frames = 100 GOPs sharded over the ranks (strong scaling)
  roofline / kernels_roofline, cpu_baseline (the reference's own --EnMultiThread path timed on this box), device_pack,
  decode, e2e (the icsp_enc program: file -> .bin + test_yuv.yuv), PSNR.
Every leg's output is checked (outside the timed region) against reference hashes or the oracle.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H = 352, 288
P = W * H
NMB = (W // 16) * (H // 16)
NFRAMES = 300
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
# un-fused fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T instructions-lanes/s; a fused multiply-add
# counts two flops (78.6 Tflop/s, the spec figure), a separate multiply or add one.  (Not in the guide; derived from the spec.)
FP64_VALU_PEAK_GOPS = 256 * 4 * 16 * 2.4
EVENT_EVERY = 4                # steps between two that carry HIP events around the dominant kernel (timed region)
SETTLE_PASSES = 100            # untimed passes (about 40 ms) before the W warmup steps of every timed leg: the GPU clock ramps
                               # up over the first tens of milliseconds of load, which would make the figure depend on W and K

# Algorithmic HBM bytes per CIF frame: every input/reference/output byte crosses once.
BYTES_I_FRAME_READ = P * 3 // 2
BYTES_I_FRAME_TOTAL = P * 3 // 2 + P * 3 // 2 + 3 * P + 10 * NMB       # + recon + int16 levels + acflag/mpm
BYTES_INTRA_LUMA_KERNEL = P + P + 2 * P + 8 * NMB                      # luma only: read P, write recon P + levels 2P + 8 B/MB
BYTES_P_FRAME_READ = 3 * P
# per P frame and kernel: k_me reads cur luma+chroma 1.5P + reference 1.5P, writes 64 B/MB; k_frame_serial moves 84 B/MB;
# k_residual reads cur 1.5P + prediction 1.5P, writes recon 1.5P + levels 3P + 8 B/MB
BYTES_P_KERNEL = {"k_me": 3 * P + 64 * NMB, "k_frame_serial": 84 * NMB, "k_residual": 3 * P + P * 3 // 2 + 3 * P + 8 * NMB}

from icspcodec_amd.workloads import CLIPS12  # noqa: E402,F401


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(a, argv):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child job.  Nothing in this process has touched the
    GPU (no HIP call, not even torch.cuda.is_available()), and the child is a fresh process, never an exec of this one."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline():
    """Reference --EnMultiThread path (oracle/_ref/icsp_ref, built from /root/reference) on a 300-frame all-I clip; falls
    back to the oracle's GOP thread pool ("port") when the reference binary did not travel.  Rank 0 at N=1 only."""
    from icspcodec_amd import clipgen
    from oracle import pyoracle as po
    cores = len(os.sched_getaffinity(0)) or 1
    clip = clipgen.synth_clip("foremanlike", NFRAMES)
    port_threads = min(cores, 64)
    t0 = time.perf_counter()
    po.encode_sequence(clip, W, H, 16, 16, 1, nthreads=port_threads)
    port_dt = time.perf_counter() - t0
    port = {"value": round(NFRAMES / port_dt, 2), "unit": "frames/s", "cores": port_threads, "kind": "port",
            "sample": f"oracle C restatement (gcc -O2 -ffp-contract=off), foremanlike 300 f all-I QP16, GOP job queue of {port_threads} threads"}
    if os.path.exists(po.REF_ENC):
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, clipgen.file_name("foremanlike", NFRAMES))
            clip.tofile(path)
            t0 = time.perf_counter()
            r1 = po.run_ref_encoder(path, NFRAMES, 16, 0, threads=0, cwd=tmp)
            dt1 = time.perf_counter() - t0
            # The reference tests Q.empty() outside its mutex (ENC:191), so large pools pop an empty queue and crash;
            # its help text documents 0-4 threads.  Try 8, then 4.
            for threads in (8, 4):
                if threads > cores:
                    continue
                # --intraPeriod 1 is the all-intra mode the thread pool accepts (period 0 divides by zero, ICSP_thread.cpp:43)
                t0 = time.perf_counter()
                r = po.run_ref_encoder(path, NFRAMES, 16, 1, threads=threads, cwd=tmp)
                dt = time.perf_counter() - t0
                if r.returncode == 0 and r1.returncode == 0:
                    return {"value": round(NFRAMES / dt, 2), "unit": "frames/s", "cores": threads, "kind": "reference",
                            "single_thread": {"value": round(NFRAMES / dt1, 2), "unit": "frames/s", "cores": 1,
                                              "sample": "reference binary, --intraPeriod 0, no thread pool (BASELINE configs[0])"},
                            "sample": f"reference binary (g++ -O2) whole process incl. file load and (single-thread) bitstream writing, "
                                      f"foremanlike 300 f all-I QP16, --intraPeriod 1 --EnMultiThread {threads} (its queue races "
                                      f"beyond a few threads); host has {cores} logical cores",
                            "port_all_cores": port}
    return port


# ------------------------------------------------------------------------------------------------ issue model
# What a vector wave-instruction costs a SIMD's issue port, by class, measured by wall clock with 2-4 waves on the SIMD
# (tools/probe_issue_waves.hip -> profiles/r06_probe_issue_waves.txt, condensed in profiles/issue_costs.json).
# fp64 add / mul / fma, conversions and the "quarter-rate" 32-bit instructions (v_sad_u8, DPP moves, v_med3, multiplies) hold the
# port about 4.3 cycles; simple integer instructions (add, and, shifts, compares, selects) 2.3 cycles in a stream of their own, but
# about 3.7 between fp64 instructions (the 50/50 mixes): `simple_int_mixed` is what the codec's kernels see, `simple_int_pure` the best case.
ISSUE_COSTS_DEFAULT = {"ns": {"fp64": 1.80, "cvt": 1.80, "other": 1.76, "simple_int_mixed": 1.55, "simple_int_pure": 0.95},
                       "simple_share_of_rest": 0.85, "cvt_share_default": 0.07}


def issue_costs():
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "issue_costs.json")))
        return {"ns": dict(ISSUE_COSTS_DEFAULT["ns"], **c.get("ns", {})), "simple_share_of_rest": c.get("simple_share_of_rest", 0.85),
                "cvt_share_default": c.get("cvt_share_default", 0.07)}
    except Exception:
        return ISSUE_COSTS_DEFAULT


def issue_seconds(total, fp64, cvt=None, n_simd=1024, best_case=False):
    """Seconds the chip's SIMDs need to ISSUE `total` vector wave-instructions of which `fp64` are fp64 add / mul / fma and `cvt`
    conversions (None: the default share); the rest split into simple integer and quarter-rate instructions by the static census of
    the kernels (profiles/issue_costs.json)."""
    c = issue_costs()
    ns = c["ns"]
    cvt = c["cvt_share_default"] * total if cvt is None else cvt
    rest = max(0.0, total - fp64 - cvt)
    simple = rest * c["simple_share_of_rest"]
    t = fp64 * ns["fp64"] + cvt * ns["cvt"] + (rest - simple) * ns["other"] + simple * (ns["simple_int_pure"] if best_case else ns["simple_int_mixed"])
    return t * 1e-9 / n_simd


# ------------------------------------------------------------------------------------------------ output
DETAIL_PATH = os.environ.get("ICSP_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
LINE_LIMIT = 8192              # bytes of the final stdout line (the driver could not parse a 21 KB line)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _finite(x):
    """JSON has no NaN / Infinity: they become null (in the detail file too)."""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as numbers, one short object per leg.
    Prose (`*_is`, notes), per-kernel counter breakdowns, the repeats and the host program's stats stay in the detail file."""
    roof = full.get("roofline") or {}
    r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_ms",
                     "algorithmic_bytes_per_launch", "launches_per_step", "whole_frame_read_frac", "whole_frame_rw_frac",
                     "valu_issue_frac", "fp64_valu_frac"))
    r.setdefault("traffic", None)
    if roof.get("chip_level"):
        r["chip_level"] = _pick(roof["chip_level"], ("achieved", "frac"))
    if roof.get("binding"):
        r["binding"] = {k: v for k, v in roof["binding"].items()
                        if (not isinstance(v, (str, dict, list)) or k == "resource") and k not in ("ceiling_fps_at_current_insts_best_case", "ceiling_fps_uniform_4_cycles")}
    cpu = full.get("cpu_baseline")
    c = None
    if cpu:
        c = _pick(cpu, ("value", "unit", "cores", "kind", "host_cores_available"))
        c["sample"] = (cpu.get("sample") or "")[:160]
        for k in ("single_thread", "port_all_cores"):
            if cpu.get(k):
                c[k] = _pick(cpu[k], ("value", "cores"))
    cfg = full.get("config") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": "foremanlike_cif 352x288 300f all-intra QP=16 per GPU (BASELINE configs[1]); steps alternate between two "
                                 "resident 300-frame batches",
                     "frames_per_step_per_gpu": cfg.get("frames_per_step_per_gpu"), "parallelism": cfg.get("parallelism")}
    out["roofline"] = r
    out["cpu_baseline"] = c
    out["parity"] = full.get("parity")
    out["psnr_y_db"] = full.get("psnr_y_db")
    # the single-clip figures beside `value`: one resident range again and again, one pass in isolation
    out["value_same_range"] = full.get("value_same_range")
    out["value_three_batches"] = full.get("value_three_batches")
    out["isolated_pass_fps"] = (full.get("isolated_pass") or {}).get("fps_per_gpu")
    out["spread_pct"] = (full.get("repeats") or {}).get("spread_pct")
    out["n_repeats"] = (full.get("repeats") or {}).get("repeats")
    ip = full.get("ippp")
    if ip:
        o = _pick(ip, ("value", "unit", "ms_per_step", "read_roofline_frac", "value_same_range", "psnr_y_db", "launches_per_step", "decode_fps"))
        o["isolated_pass_fps"] = (ip.get("isolated_pass") or {}).get("fps")
        o["workload"] = "stefanlike_cif 300f --intraPeriod 10 QP=8 per GPU (BASELINE configs[2])"
        out["ippp"] = o
    c4 = full.get("config4")
    if c4:
        o = _pick(c4, ("value", "unit", "ms_per_pass", "frames", "scaling", "read_roofline_frac", "recon_equals_reference"))
        o["workload"] = "12 CIF clips, 3390 f, --intraPeriod 10 QP=16, GOP-sharded (BASELINE configs[3])"
        o["gops_per_rank"] = (c4.get("regime") or {}).get("gops_per_rank")
        iss = c4.get("issue") or {}
        o.update(_pick(iss, ("valu_issue_frac", "traffic_over_algorithmic")))
        ai = c4.get("all_intra_loaded") or {}
        a_ = _pick(ai, ("value", "ms_per_pass", "read_roofline_frac", "rw_roofline_frac"))
        a_.update(_pick(ai.get("issue") or {}, ("valu_issue_frac", "traffic_over_algorithmic")))
        o["all_intra_loaded"] = a_
        if c4.get("per_rank_projection"):
            o["per_rank_projection"] = c4["per_rank_projection"]
        out["config4"] = o
    c5 = full.get("config5")
    if c5:
        o = _pick(c5, ("value", "unit", "ms_per_pass", "frames", "scaling", "read_roofline_frac", "recon_equals_oracle", "cif_equivalent_fps"))
        o["workload"] = "1920x1088 3000 f --intraPeriod 30 QP=16, GOP-sharded (BASELINE configs[4])"
        o["gops_per_rank"] = (c5.get("regime") or {}).get("gops_per_rank")
        o.update(_pick(c5.get("issue") or {}, ("valu_issue_frac",)))
        if c5.get("per_rank_projection"):
            o["per_rank_projection"] = c5["per_rank_projection"]
        out["config5"] = o
    e = full.get("e2e")
    if e:
        o = _pick(e, ("rc", "wall_fps_incl_process_start_and_hip_init", "bin_equals_reference", "recon_equals_reference"))
        for k in ("long_3000f_ippp", "long_3000f_all_intra"):
            if e.get(k):
                o[k + "_fps_excl_init"] = (e[k].get("stats") or {}).get("e2e_fps_excl_init")
        out["e2e"] = o
    pc = full.get("pcie_inclusive")
    if pc:
        out["pcie_inclusive"] = {k: v for k, v in pc.items() if k.endswith("_fps")}
    out["small_ranges"] = {k: v for k, v in (full.get("small_ranges") or {}).items() if k.endswith("_fps")}
    out["device_pack"] = _pick(full.get("device_pack") or {}, ("kernels_ms", "upload_encode_pack_fps"))
    out["decode_fps"] = (full.get("decode") or {}).get("all_intra_fps")
    out["regime"] = _pick(full.get("regime") or {}, ("intra_lanes_per_block", "intra_waves", "whole", "gop_groups"))
    out["detail"] = os.path.basename(DETAIL_PATH)
    return out


def emit(full):
    """Everything into the detail file (path announced on a line of its own, which does not start with '{'), then the ONE short
    JSON line, last on stdout."""
    full = _finite(full)
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(full, f, indent=1)
        print("bench detail: " + DETAIL_PATH)
    except OSError as e:
        print("bench detail: not written (%s)" % e)
    text = json.dumps(compact_line(full), separators=(",", ":"), allow_nan=False)
    if len(text) >= LINE_LIMIT:          # never hand the driver a line it cannot read: drop the secondary legs' extras first
        c = compact_line(full)
        for k in ("small_ranges", "device_pack", "pcie_inclusive", "e2e", "regime", "decode_fps"):
            c.pop(k, None)
        text = json.dumps(c, separators=(",", ":"), allow_nan=False)
    assert len(text) < LINE_LIMIT, len(text)
    sys.stdout.flush()
    print(text, flush=True)



# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="how often the K-step timed region is repeated (value = the median)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--legs", default="ippp,config4,config5,e2e", help="secondary legs to run (comma list; '' = none)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a, sys.argv[1:]))
    legs = {x for x in a.legs.split(",") if x}

    import torch
    import torch.distributed as dist
    from icspcodec_amd import capi, clipgen, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test hooks (tests/test_gpu_bench.py runs the N>1 code path with two ranks on a one-GPU box): the driver never sets them
    backend = os.environ.get("ICSP_BENCH_BACKEND", "nccl")
    if "ICSP_BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["ICSP_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    n_cu_dev = int(torch.cuda.get_device_properties(local).multi_processor_count)
    # this rank's host thread (and the threads it starts from here on) onto the NUMA node its GPU hangs off: a no-op on one node
    _lib = capi.load()
    import ctypes as _C
    import contextlib
    _bound = _C.c_int(0)
    _node = _lib.icsp_device_numa_node(local)
    _aff_all = os.sched_getaffinity(0)                 # before the binding: what the CPU-side legs (baseline, oracle checks) may use
    _lib.icsp_bind_thread_to_node(_node, _C.byref(_bound))
    _aff_bound = os.sched_getaffinity(0)

    @contextlib.contextmanager
    def all_cores():
        """The CPU baseline and the oracle's thread pools size themselves by the host's cores; run them on every core this
        process was given, not on the GPU's NUMA node alone, and go back to the node afterwards."""
        try:
            os.sched_setaffinity(0, _aff_all)
        except OSError:
            pass
        try:
            yield
        finally:
            try:
                os.sched_setaffinity(0, _aff_bound)
            except OSError:
                pass
    # The device's four compute streams before anything else of this process creates streams (RCCL does, below): which hardware
    # queues they get depends on the order of creation (profiles/r05_stream_order.txt).  A throw-away context
    # encodes two tiny IPPP ranges in turn -- that makes all four -- and is closed; the legs' contexts inherit its streams.
    _rs = capi.Encoder(W, H, 8, 8, 2, device=local, max_frames=8)
    _rs.upload(np.zeros((8, W * H * 3 // 2), np.uint8))
    for _k in range(4):
        _rs.encode_resident((_k & 1) * 4, 4)
    _rs.sync()
    _rs.close()
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))      # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(x, op):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())
    ranks_seen = int(reduce(1.0, dist.ReduceOp.SUM)) if world > 1 else 1

    # counters of the round's rocprofv3 --pmc passes over the same workloads (tools/profile_round.sh -> profiles/traffic.json)
    try:
        _tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        _tj = {}
    ISSUE_PEAK = 256 * 4 * 2.4e9 / 4                     # wave-instructions/s: 256 CUs x 4 SIMDs, one per 4 cycles at 2.4 GHz

    def leg_issue(leg, sec_per_pass, ranks=1, choice=None):
        """What binds a loaded leg: the vector instructions of ONE pass of the leg (SQ_INSTS_VALU summed over the pass's
        launches, rocprofv3 --pmc on tools/leg_workload.py <leg>, whole batch on one GPU) over this run's time per pass, as a share
        of the chip's vector issue slots; the fp64 share; how much of their lifetime the waves spent waiting.  None without
        a counter file of the leg.  (N ranks: each holds 1/N of the batch.)"""
        e = (_tj.get("legs") or {}).get(leg)
        if not e or not sec_per_pass:
            return None
        # the counters are those of the WHOLE batch on one GPU: a shard of it runs other kernel forms at other loads, so the
        # figure is given for one rank only, and only while this run made the choices the counted run made
        if ranks != 1:
            return {"skipped": "counters in profiles/traffic.json are of the whole batch on one GPU; this run shards it over %d ranks" % ranks}
        if choice is not None and e.get("choice") and any(choice.get(k) != v for k, v in e["choice"].items()):
            return {"skipped": "this run chose other kernel forms than the counted run", "counted_choice": e["choice"], "this_choice": choice}
        vi, f64 = e.get("valu_insts_per_pass", 0), e.get("fp64_valu_insts_per_pass", 0)
        out = {"valu_issue_frac": round(issue_seconds(vi / ranks, f64 / ranks, n_simd=4 * n_cu_dev) / sec_per_pass, 4),
               "valu_issue_frac_uniform_4_cycles": round(vi / ranks / sec_per_pass / ISSUE_PEAK, 4),
               "fp64_valu_frac": round(f64 / ranks * 64 / sec_per_pass / 1e9 / FP64_VALU_PEAK_GOPS, 4),
               "waiting_share_of_wave_cycles": e.get("waiting_share_of_wave_cycles"),
               "valu_insts_per_pass": int(vi), "hbm_bytes_per_pass": e.get("hbm_bytes_per_pass"),
               "traffic_over_algorithmic": e.get("traffic_over_algorithmic"),
               "by_kernel": e.get("by_kernel"),
               "source": e.get("source"),
               "is": "time the chip's SIMDs need to issue the vector wave-instructions of one pass, at the per-class costs measured by "
                     "tools/probe_issue_waves.hip (profiles/issue_costs.json), over this run's time per pass: the share of the chip's vector "
                     "issue time the leg fills -- the ceiling that binds these kernels (DESIGN.md section 5); *_uniform_4_cycles: rounds "
                     "4-5's model, every instruction 4 cycles at 2.4 GHz"}
        return out

    def timed(enc, n, steps, warmup, dominant, alternate=True, ranges=2):
        """The K-step timed region, `a.repeats` times over (each bracketed by barrier + synchronize; the list of their
        durations comes back, max over ranks each): steps alternating between the resident batches at slots [0, n) and
        [n, 2n) (`ranges` batches in rotation) -- or, alternate=False, the batch at [0, n) again and again -- with HIP events only around the dominant kernel
        (none if it is None); then four untimed passes with events on every kernel."""
        step_no = [0]

        def one_step():
            enc.encode_resident((step_no[0] % ranges) * n if alternate else 0, n)
            step_no[0] += 1
        for _ in range(SETTLE_PASSES):              # fixed settle (clock ramp, instruction and TLB warm-up): independent of W
            one_step()
        enc.sync()
        for _ in range(warmup):
            one_step()
        enc.sync()
        flag = 0
        if dominant:
            enc.profile(True, only=[dominant])          # creates the events, clears the sums
            flag = 1 << (capi.KERNELS.index(dominant) + 1)
            enc.lib.icsp_profile_enable(enc.ctx, 0)
        dts = []
        for _rep in range(max(1, a.repeats)):
            barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                # the two event records around the dominant kernel cost the stream about 15 us a step (5 % of this one): every
                # EVENT_EVERY-th step carries them, the average launch duration is over those
                ev = flag and i % EVENT_EVERY == 0
                if ev:
                    enc.lib.icsp_profile_enable(enc.ctx, flag)
                one_step()
                if ev:
                    enc.lib.icsp_profile_enable(enc.ctx, 0)
            enc.sync()
            barrier()
            dts.append(reduce(time.perf_counter() - t0, dist.ReduceOp.MAX))
        dom_ms, dom_n = enc.profile_get()[dominant] if dominant else (0.0, 0)
        enc.profile(True)
        for _ in range(4):
            one_step()
        enc.sync()
        prof = {k: (v[0] / 4.0, v[1] // 4) for k, v in enc.profile_get().items()}      # (ms per pass, launches per pass)
        enc.profile(False)
        return dts, prof, (dom_ms, dom_n)

    def med(xs):
        return sorted(xs)[len(xs) // 2]

    def spread(dts, frames_per_region):
        """the repeats of a timed region as rates: median, fastest, slowest"""
        r = sorted(frames_per_region / d for d in dts)
        return {"repeats": len(r), "median": round(r[len(r) // 2], 1), "min": round(r[0], 1), "max": round(r[-1], 1),
                "spread_pct": round(100.0 * (r[-1] - r[0]) / r[len(r) // 2], 2)}

    def timed_passes(enc, n, passes):
        """Big batches: warm passes for at least 40 ms (two at the least: the GPU clock ramps), then `passes` timed ones
        between barriers; max over ranks."""
        t_w = time.perf_counter()
        k = 0
        while k < 2 or (time.perf_counter() - t_w < 0.04 and k < 50):
            enc.encode_resident(0, n)
            enc.sync()
            k += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(passes):
            enc.encode_resident(0, n)
        enc.sync()
        barrier()
        return reduce(time.perf_counter() - t0, dist.ReduceOp.MAX) / passes

    def isolated_pass_ms(enc, n, reps=12):
        """One pass at a time, the host waiting for each: no pass overlaps the one before (the timed steps do: consecutive
        passes over the same resident range follow each other part by part / group by group).  Median of `reps`."""
        ts = []
        for _ in range(reps):
            enc.sync()
            t0 = time.perf_counter()
            enc.encode_resident(0, n)
            enc.sync()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    # ---- primary: configs[1] all-intra QP16, each rank its own frames of the synthetic sequence: batch A in slots [0, 300),
    #      batch B (the next 300 frames of the sequence) in slots [300, 600); the steps alternate between them
    from oracle import pyoracle as po
    ncore = min(os.cpu_count() or 1, 64)
    clip = clipgen.synth_clip("foremanlike", NFRAMES, first_frame=rank * 2 * NFRAMES)
    clip_b = clipgen.synth_clip("foremanlike", NFRAMES, first_frame=rank * 2 * NFRAMES + NFRAMES)
    enc = capi.Encoder(W, H, 16, 16, 0, device=local, max_frames=3 * NFRAMES)
    enc.upload(clip, first=0)
    enc.upload(clip_b, first=NFRAMES)
    dts_ai, prof, (ms_ai, n_ai) = timed(enc, NFRAMES, a.steps, a.warmup, "k_intra_luma")
    dt = med(dts_ai)
    choice_ai = enc.last_choice()
    # beside `value`, never instead of it (its regime stays two batches, like for like since round 3): THREE resident batches in
    # rotation -- a host with a ring of three chunk slots.  The library then keeps three batches in flight on three chain streams
    # (round 5), where a batch that alternates with one other must follow its own previous pass.
    enc.upload(clipgen.synth_clip("foremanlike", NFRAMES, first_frame=rank * 2 * NFRAMES + 2 * NFRAMES), first=2 * NFRAMES)
    dts_r3, _, _ = timed(enc, NFRAMES, a.steps, a.warmup, None, ranges=3)
    choice_r3 = enc.last_choice()
    # the regime of rounds 1 and 2, kept beside `value` so that round-over-round figures stay like for like: ONE
    # resident 300-frame batch encoded again and again (the library then runs it in two parts on two streams)
    dts_same, _, (ms_same, n_same) = timed(enc, NFRAMES, a.steps, a.warmup, "k_intra_luma", alternate=False)
    choice_same = enc.last_choice()
    # small ranges (secondary, never `value`): the same 600 resident frames as four ranges of 150, handed over
    # one by one (A, B, C, D, A ...) and as two LISTS of two ranges in turn (icsp_encode_resident_many: one launch per list)
    def small_ranges():
        q = NFRAMES // 2
        out = {}
        for key, lists in (("one_by_one_fps", None), ("two_lists_of_two_fps", [[(0, q), (2 * q, q)], [(q, q), (3 * q, q)]])):
            def call(k):
                if lists:
                    enc.encode_resident_many(lists[k & 1])
                else:
                    enc.encode_resident((k & 3) * q, q)
            ncall = 200 if lists else 400
            for k in range(40):
                call(k)
            enc.sync()
            best = 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                for k in range(ncall):
                    call(k)
                enc.sync()
                best = max(best, ncall * (2 * q if lists else q) / (time.perf_counter() - t0))
            out[key] = round(best * world, 1)
        out["is"] = ("four resident ranges of 150 frames (the same 600 frames as `value`): one icsp_encode_resident per range in rotation, and two "
                     "lists of two ranges each through icsp_encode_resident_many in turn (one luma and one chroma launch per list); best of three runs")
        return out
    small = small_ranges()
    enc.encode_resident(0, NFRAMES)
    enc.encode_resident(NFRAMES, NFRAMES)              # (both batches' results are checked below)
    enc.sync()
    recon = enc.download(0, NFRAMES, what=("recon",))["recon"]
    recon_b = enc.download(NFRAMES, NFRAMES, what=("recon",))["recon"]
    iso_ai = isolated_pass_ms(enc, NFRAMES)
    psnr_ai = clipgen.psnr_y(clip, recon, W, H)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "streams.json")))
    parity = {}
    if rank == 0:       # rank 0's batch A is the clip the reference CLI was run on: same recon bytes, same .bin; B against the oracle
        ref = next(s for s in golden if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
        parity["configs1_recon_sha_equals_reference"] = hashlib.sha256(recon.tobytes()).hexdigest() == ref["recon_sha256"]
        parity["configs1_bin_sha_equals_reference"] = hashlib.sha256(enc.pack_bitstream(0, NFRAMES)).hexdigest() == ref["bin_sha256"]
        with all_cores():
            parity["configs1_batch_b_recon_equals_oracle"] = bool(np.array_equal(recon_b, po.encode_sequence(clip_b, W, H, 16, 16, 1, nthreads=ncore)["recon"]))
    del recon_b
    # PCIe-inclusive (host buffers in, host results out through the one-call boundary icsp_encode_gop) — reported, never `value`.
    # The caller's arrays exist before the call (the reference allocates its frame store once per clip, ENC:247-283), either
    # plain memory or pinned (icsp_host_alloc, what INTEGRATION.md's binding uses); median of five calls after two warm ones.
    def gop_rates(e, src_clip, period_tag):
        nmb_ = NMB
        shapes = dict(levels=((NFRAMES, nmb_, 6, 64), np.int16), acflag=((NFRAMES, nmb_, 6), np.uint8), mpm=((NFRAMES, nmb_, 4), np.uint8),
                      mvd=((NFRAMES, nmb_, 2), np.int8), recon=((NFRAMES, W * H * 3 // 2), np.uint8))
        res = {}
        nb = e.lib.icsp_bitstream_bound(_C.byref(e.params), NFRAMES)
        for mem in ("pageable", "pinned"):
            try:
                if mem == "pinned":
                    src = capi.host_alloc_array(src_clip.shape, np.uint8)
                    src[:] = src_clip
                    out = {k: capi.host_alloc_array(sh, d) for k, (sh, d) in shapes.items()}
                    body = capi.host_alloc_array((nb,), np.uint8)
                else:
                    src = src_clip.copy()
                    out = {k: np.zeros(sh, d) for k, (sh, d) in shapes.items()}
                    body = np.zeros(nb, np.uint8)
            except Exception:
                continue
            for fn, key in ((lambda: e.encode(src, out=out), mem + "_fps"), (lambda: e.encode_packed(src, recon=out["recon"], body=body), "packed_" + mem + "_fps"),
                            (lambda: e.encode_packed(src, recon=None, body=body), "packed_norecon_" + mem + "_fps")):
                for _ in range(2):
                    fn()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                res[key] = round(NFRAMES / sorted(ts)[2], 1)
            if mem == "pinned":
                for arr in list(out.values()) + [src, body]:
                    capi.host_free_array(arr)
        res["is"] = ("icsp_encode_gop on 300 host frames, " + period_tag + ": frames up, levels + flags + vectors + reconstruction down (613 KB per frame), "
                     "pipelined in chunks inside the call; packed_*: icsp_encode_gop_packed, reconstruction + packed body down (165 KB per frame); "
                     "packed_norecon_*: the same with recon = NULL, frames up and only the bits down (what makebitstream needs, ENC:222,242)")
        return res
    pcie = gop_rates(enc, clip, "all-intra QP16")
    pcie_dt = NFRAMES / max(pcie.get("pinned_fps", pcie.get("pageable_fps", 1.0)), 1.0)
    # the same with the body packed on the device, so that only the bits come back (icsp_pack_bits)
    hostbuf = np.empty(64 << 20, np.uint8)
    enc.pack_bits(0, NFRAMES, hostbuf)
    enc.profile(True, only=["k_pack"])
    t0 = time.perf_counter()
    enc.upload(clip)
    enc.encode_resident(0, NFRAMES)
    body, nbits = enc.pack_bits(0, NFRAMES, hostbuf)
    e2e_dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        enc.pack_bits(0, NFRAMES, hostbuf)
    pack_dt = (time.perf_counter() - t0) / 3
    pack_ms = enc.profile_get()["k_pack"]
    enc.profile(False)

    def decode_fps(e):
        """f4: decoder reconstruction of the resident syntax (overwrites the recon planes; run after they were used)."""
        for _ in range(2):
            e.decode_resident(0, NFRAMES)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.decode_resident(0, NFRAMES)
        e.sync()
        return NFRAMES * 5 / (time.perf_counter() - t0)
    dec_ai_fps = decode_fps(enc)
    enc.close()
    fps = world * NFRAMES * a.steps / dt
    kern_ms = ms_ai / max(n_ai, 1)
    # A step launches the luma kernel in `lps` parts on as many streams (more frames than CUs: icsp_sched.cpp, encode_range),
    # and the launches of consecutive steps (independent batches) run side by side as well, so a launch's own duration says
    # little about the chip.  Chip-level figure: the kernel's algorithmic bytes of a step over the step's share of
    # the timed region -- in steady state the span of a step's launches -- i.e. bytes per step / ms_per_step.  The per-launch
    # figure of the contract (bytes of one launch / its average duration under HIP events) is kept beside it.
    n_ev_steps = max(1, a.repeats) * ((a.steps + EVENT_EVERY - 1) // EVENT_EVERY)     # timed steps that carried events
    lps = max(1, round(n_ai / max(1, n_ev_steps)))
    step_ms = dt / a.steps * 1e3
    achieved = BYTES_INTRA_LUMA_KERNEL * NFRAMES / (step_ms * 1e-3) / 1e9
    single_launch = BYTES_INTRA_LUMA_KERNEL * NFRAMES / lps / (kern_ms * 1e-3) / 1e9 if n_ai else 0.0

    # ---- configs[2] IPPP, stefanlike --intraPeriod 10 QP8 (ME + MC path), weak like the primary
    ippp = None
    if "ippp" in legs:
        clip2 = clipgen.synth_clip("stefanlike", NFRAMES, first_frame=rank * 2 * NFRAMES)
        clip2_b = clipgen.synth_clip("stefanlike", NFRAMES, first_frame=rank * 2 * NFRAMES + NFRAMES)
        enc2 = capi.Encoder(W, H, 8, 8, 10, device=local, max_frames=2 * NFRAMES)
        enc2.upload(clip2, first=0)
        enc2.upload(clip2_b, first=NFRAMES)
        steps2 = max(2, a.steps)
        dts_ip, prof2, _ = timed(enc2, NFRAMES, steps2, a.warmup, None)      # no events inside this timed region
        dt2 = med(dts_ip)
        choice_ip = enc2.last_choice()
        dts_ip_same, _, _ = timed(enc2, NFRAMES, steps2, a.warmup, None, alternate=False)
        enc2.encode_resident(NFRAMES, NFRAMES)
        enc2.sync()
        recon2 = enc2.download(0, NFRAMES, what=("recon",))["recon"]
        recon2_b = enc2.download(NFRAMES, NFRAMES, what=("recon",))["recon"]
        iso_ip = isolated_pass_ms(enc2, NFRAMES)
        psnr_ip = clipgen.psnr_y(clip2, recon2, W, H)
        if rank == 0:
            ref = next(s for s in golden if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("stefanlike", 300, 8, 10))
            parity["configs2_recon_sha_equals_reference"] = hashlib.sha256(recon2.tobytes()).hexdigest() == ref["recon_sha256"]
            with all_cores():
                parity["configs2_batch_b_recon_equals_oracle"] = bool(np.array_equal(recon2_b, po.encode_sequence(clip2_b, W, H, 8, 8, 10, nthreads=ncore)["recon"]))
        del recon2_b
        dec_ip_fps = decode_fps(enc2)
        enc2.close()
        fps2 = world * NFRAMES * steps2 / dt2
        read_mean_ip = (BYTES_I_FRAME_READ + 9 * BYTES_P_FRAME_READ) / 10.0
        kr = {}
        for k, per_frame in BYTES_P_KERNEL.items():
            ms, nl = prof2.get(k, (0.0, 0))
            if nl:
                frames_per_launch = 270.0 / nl                 # 270 P frames per pass, spread over the launches of that kernel
                gbs = per_frame * frames_per_launch / (ms / nl * 1e-3) / 1e9
                kr[k] = {"avg_launch_us": round(ms / nl * 1e3, 2), "frames_per_launch": round(frames_per_launch, 1),
                         "algorithmic_bytes_per_launch": int(per_frame * frames_per_launch), "achieved_GBps": round(gbs, 1),
                         "hbm_frac": round(gbs / HBM_PEAK_GBS, 5)}
        # counter-measured HBM-side bytes of the same launches (profiles/traffic.json, tools/profile_round.sh), where committed
        try:
            tk = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernels"]
            for mine, theirs in (("k_me", "k_me_false@"), ("k_residual", "k_residual8@"), ("k_frame_serial", "k_serial_fused@")):
                for name, e in tk.items():
                    if name.startswith(theirs) and e.get("frames_per_launch", 15) == 15 and e.get("launches_seen", 0) >= 27 and mine in kr:
                        # (the counters are those of a 15-frame launch of the same kernel; a launch here carries frames_per_launch)
                        if e.get("hbm_bytes_per_launch"):
                            kr[mine]["traffic_bytes_per_launch"] = int(e["hbm_bytes_per_launch"] * kr[mine]["frames_per_launch"] / 15.0)
                        if "traffic_over_algorithmic" in e:
                            kr[mine]["traffic_over_algorithmic"] = e["traffic_over_algorithmic"]
        except Exception:
            pass
        ippp = {"workload": "stefanlike_cif 300f, --intraPeriod 10, QP=8 (BASELINE configs[2]), per GPU; steps alternate between two "
                            "disjoint resident 300-frame batches (frames 0-299 and 300-599 of the sequence)", "value": round(fps2, 1),
                "unit": "frames/s", "ms_per_step": round(dt2 / steps2 * 1e3, 4), "psnr_y_db": round(psnr_ip, 4),
                "value_is": "median of the repeats of the K-step region", "repeats": spread(dts_ip, world * NFRAMES * steps2),
                "value_same_range": round(world * NFRAMES * steps2 / med(dts_ip_same), 1),
                "same_range_repeats": spread(dts_ip_same, world * NFRAMES * steps2),
                "isolated_pass": {"ms": round(iso_ip, 4), "fps": round(NFRAMES / iso_ip * 1e3, 1),
                                  "note": "one pass, host waits before and after: nothing runs beside it"},
                "regime": dict(choice_ip, gops_per_rank_per_step=NFRAMES // 10, frames_in_flight_per_cu=round(2 * NFRAMES / 256, 2)),
                "read_roofline_frac": round(fps2 / world * read_mean_ip / 1e9 / HBM_PEAK_GBS, 5),
                "kernels_ms_per_step": {k: round(v[0], 4) for k, v in prof2.items() if v[1]},
                "launches_per_step": {k: v[1] for k, v in prof2.items() if v[1]},
                "kernels_roofline": kr, "decode_fps": round(dec_ip_fps, 1),
                "kernels_note": "HIP-event time summed over launches (the events themselves cost a few us per launch); the GOP "
                                "groups' P-step chains and the I-frame chroma run on concurrent streams, so the sum exceeds ms_per_step"}

    # ---- configs[3]: the twelve clips as ONE batch of 339 closed GOPs, sharded over the ranks (strong scaling)
    config4 = None
    if "config4" in legs:
        from icspcodec_amd import workloads
        batch, mine, ntot = workloads.clips12_shard(rank, world)
        nloc = batch.shape[0]
        enc4 = capi.Encoder(W, H, 16, 16, 10, device=local, max_frames=nloc)
        enc4.upload(batch)
        sec = timed_passes(enc4, nloc, 5)
        choice4 = enc4.last_choice()
        iso4 = isolated_pass_ms(enc4, nloc, reps=5)
        rec4 = enc4.download(0, nloc, what=("recon",))["recon"]
        ok4 = True
        if world == 1:        # whole clips on this rank: the reference CLI's recon hashes (tests/golden/streams.json)
            pos = 0
            for name in CLIPS12:
                n = clipgen.CLIP_CLASSES[name]["nframes"]
                ref = next(s for s in golden if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == (name, n, 16, 10) and "bin_sha256" in s)
                ok4 &= hashlib.sha256(rec4[pos: pos + n].tobytes()).hexdigest() == ref["recon_sha256"]
                pos += n
        else:                 # a rank holds runs of GOPs: its first and last GOP against the oracle
            with all_cores():
                for sl in (slice(0, mine[0][2]), slice(nloc - mine[-1][2], nloc)):
                    ok4 &= np.array_equal(rec4[sl], po.encode_sequence(batch[sl], W, H, 16, 16, 10)["recon"])
        ok4 = reduce(1.0 if ok4 else 0.0, dist.ReduceOp.MIN) == 1.0 if world > 1 else ok4
        enc4.close()
        proj4 = None
        if world == 1:
            # What ONE rank of an N-way split holds, run on this device alone: N x its rate is what N GPUs would give if they did not
            # disturb each other at all -- a PROJECTION from one device, not a measurement of N
            proj4 = {}
            for nr in (2, 4, 8):
                sb, smine, _ = workloads.clips12_shard(0, nr)
                es = capi.Encoder(W, H, 16, 16, 10, device=local, max_frames=sb.shape[0])
                es.upload(sb)
                ssec = timed_passes(es, sb.shape[0], 20)
                es.close()
                proj4[str(nr)] = {"gops": len(smine), "frames": int(sb.shape[0]), "rank_fps": round(sb.shape[0] / ssec, 1),
                                  "fps_x_ranks": round(nr * sb.shape[0] / ssec, 1), "implied_efficiency": round(sb.shape[0] / ssec / (ntot / sec), 4)}
                del sb
        # the same frames all-intra: every CU busy (the throughput regime of the intra kernels)
        enc4i = capi.Encoder(W, H, 16, 16, 0, device=local, max_frames=nloc)
        enc4i.upload(batch)
        sec_i = timed_passes(enc4i, nloc, 5)
        choice4i = enc4i.last_choice()
        enc4i.close()
        config4 = {"workload": "12 CIF clips (11 x 300 f + 1 x 90 f = 3390 frames, 339 closed GOPs), --intraPeriod 10, QP=16, one "
                               "batch sharded by GOP over the ranks (BASELINE configs[3])", "scaling": "strong",
                   "value": round(ntot / sec, 1), "unit": "frames/s", "ms_per_pass": round(sec * 1e3, 3), "frames": ntot,
                   "recon_equals_reference": bool(ok4),
                   # what decides a rank's efficiency on the strong-scaling legs: how much of the batch it holds
                   "regime": dict(choice4, gops_per_rank=len(mine), frames_per_rank=nloc, p_frames_per_step_per_cu=round(len(mine) / 256, 3),
                                  i_frames_per_cu=round(len(mine) / 256, 3),
                                  isolated_pass_ms_this_rank=round(iso4, 3), isolated_pass_fps_this_rank=round(nloc / iso4 * 1e3, 1),
                                  note="fewer GOPs per rank = fewer frames per launch: from the loaded regime (339 GOPs on one GPU) "
                                       "towards the latency regime of configs[2] (30 GOPs); a rank's own isolated pass is listed so that "
                                       "an N-rank line explains its efficiency"),
                   "read_roofline_frac": round(ntot / sec * (BYTES_I_FRAME_READ + 9 * BYTES_P_FRAME_READ) / 10.0 / 1e9 / HBM_PEAK_GBS / world, 5),
                   "issue": leg_issue("config4", sec, world, choice4),
                   "per_rank_projection": proj4,
                   "per_rank_projection_is": "projection, not a measurement: rank 0's share of an N-way GOP split (N = 2, 4, 8) encoded on THIS one "
                                             "device alone (passes back to back, like `value` of this leg), rate x N; implied_efficiency = that / (N x "
                                             "this leg's one-GPU value).  No run on N devices stands behind it",
                   "all_intra_loaded": {"value": round(ntot / sec_i, 1), "unit": "frames/s", "ms_per_pass": round(sec_i * 1e3, 3),
                                        "issue": leg_issue("config4_allintra", sec_i, world, choice4i),
                                        "regime": dict(choice4i, frames_per_rank=nloc, frames_per_cu=round(nloc / 256, 2)),
                                        "read_roofline_frac": round(ntot / sec_i * BYTES_I_FRAME_READ / 1e9 / HBM_PEAK_GBS / world, 5),
                                        "rw_roofline_frac": round(ntot / sec_i * BYTES_I_FRAME_TOTAL / 1e9 / HBM_PEAK_GBS / world, 5)}}
        del batch, rec4

    # ---- configs[4]: 1920x1088, --intraPeriod 30, 3000 frames = 100 GOPs sharded over the ranks (strong scaling)
    config5 = None
    if "config5" in legs:
        from icspcodec_amd import workloads
        w5, h5, L5, ngop5 = workloads.HD_W, workloads.HD_H, workloads.HD_PERIOD, workloads.HD_GOPS
        srcs = workloads.HD_SRCS
        # "CIF-tiled macroblock grid": a 1088p frame is the CIF frame tiled; GOP g shows clip (g mod 4), so four distinct GOPs
        g_lo, g_n, gops = workloads.hd_shard(rank, world)
        enc5 = capi.Encoder(w5, h5, 16, 16, L5, device=local, max_frames=g_n * L5)
        for g in range(g_n):
            enc5.upload(gops[srcs[(g_lo + g) % 4]], first=g * L5)
        sec5 = timed_passes(enc5, g_n * L5, 2)
        choice5 = enc5.last_choice()
        iso5 = isolated_pass_ms(enc5, g_n * L5, reps=3)
        # check (outside the timed region): one resident GOP of each distinct content against the oracle's GOP thread pool
        first_of = {}
        for g in range(g_n):
            first_of.setdefault((g_lo + g) % 4, g)
        check = sorted(first_of.items())[: (4 if world == 1 else 1)]
        with all_cores():
            want = po.encode_sequence(np.concatenate([gops[srcs[k]] for k, _ in check]), w5, h5, 16, 16, L5, nthreads=len(check))["recon"]
        ok5 = True
        for j, (k, g) in enumerate(check):
            ok5 &= np.array_equal(enc5.download(g * L5, L5, what=("recon",))["recon"], want[j * L5: (j + 1) * L5])
        ok5 = reduce(1.0 if ok5 else 0.0, dist.ReduceOp.MIN) == 1.0 if world > 1 else ok5
        enc5.close()
        proj5 = None
        if world == 1:
            proj5 = {}
            for nr in (2, 4, 8):
                s_lo, s_n = 0, (ngop5 + nr - 1) // nr            # rank 0's share (hd_shard: the first ranks take the remainder)
                es = capi.Encoder(w5, h5, 16, 16, L5, device=local, max_frames=s_n * L5)
                for g in range(s_n):
                    es.upload(gops[srcs[(s_lo + g) % 4]], first=g * L5)
                ssec = timed_passes(es, s_n * L5, 3)
                es.close()
                proj5[str(nr)] = {"gops": s_n, "frames": s_n * L5, "rank_fps": round(s_n * L5 / ssec, 1), "fps_x_ranks": round(nr * s_n * L5 / ssec, 1),
                                  "implied_efficiency": round(s_n * L5 / ssec / (ngop5 * L5 / sec5), 4)}
        n5 = ngop5 * L5
        rd5 = (w5 * h5 * 3 // 2 + 29 * 3 * w5 * h5) / 30.0
        config5 = {"workload": "1920x1088 (CIF-tiled macroblock grid), 3000 frames = 100 closed GOPs of 30, QP=16, sharded by GOP over "
                               "the ranks (BASELINE configs[4]; 1080 is not a multiple of 16)", "scaling": "strong",
                   "value": round(n5 / sec5, 1), "unit": "frames/s", "ms_per_pass": round(sec5 * 1e3, 2), "frames": n5,
                   "cif_equivalent_fps": round(n5 / sec5 * (w5 * h5) / P, 1), "recon_equals_oracle": bool(ok5),
                   "regime": dict(choice5, gops_per_rank=g_n, frames_per_rank=g_n * L5, macroblocks_per_p_step_per_cu=round(g_n * 8160 / 256, 1),
                                  isolated_pass_ms_this_rank=round(iso5, 2), isolated_pass_fps_this_rank=round(g_n * L5 / iso5 * 1e3, 1)),
                   "issue": leg_issue("config5", sec5, world, choice5),
                   "per_rank_projection": proj5,
                   "per_rank_projection_is": "projection, not a measurement (see config4.per_rank_projection_is): 50 / 25 / 13 GOPs of 1088p on this one device",
                   "read_roofline_frac": round(n5 / sec5 * rd5 / 1e9 / HBM_PEAK_GBS / world, 5)}
        del gops

    # ---- the host program end to end: file -> .bin + test_yuv.yuv (rank 0, N = 1 only)
    e2e = None
    enc_bin = os.path.join(ROOT, "icspcodec_amd", "icsp_enc")
    if "e2e" in legs and world == 1 and os.path.exists(enc_bin):
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            path = os.path.join(tmp, clipgen.file_name("foremanlike", NFRAMES))
            clip.tofile(path)
            t0 = time.perf_counter()
            # relative input name: the output prefix is the input PATH up to its first '_' (encoder_main.cpp:10-17), and a
            # temporary directory's name may contain one
            def run_enc(args):
                """icsp_enc as a child process, bounded: a stuck run is reported, never waited for."""
                try:
                    return subprocess.run([enc_bin] + args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
                except subprocess.TimeoutExpired as e:
                    return subprocess.CompletedProcess(e.cmd, "timeout", e.stdout or b"", None)
            # three runs, a second apart (hsa_init takes 0.15-0.25 s instead of 0.06 when the previous GPU process exited a moment
            # ago).  The FIRST is the first run of the program on this box: before round 4 it paid 0.14 s of page
            # faults into the HIP runtime's libraries from a cold disk (tools/cold_cache.sh), which icsp_enc now reads ahead.
            walls = []
            for _ in range(3):
                time.sleep(1.0)
                t0 = time.perf_counter()
                r = run_enc(["-i", os.path.basename(path), "-n", str(NFRAMES), "-q", "16", "--intraPeriod", "0", "--stats"])
                walls.append(time.perf_counter() - t0)
                if _ == 0:
                    first_out = r.stdout.decode(errors="replace")
            wall = sorted(walls)[1]
            out = r.stdout.decode(errors="replace")
            e2e = {"workload": "icsp_enc: foremanlike 300 f file -> .bin + test_yuv.yuv (tmpfs), all-intra QP16", "rc": r.returncode,
                   "wall_fps_incl_process_start_and_hip_init": round(NFRAMES / wall, 1),
                   "wall_is": "median of three runs of the program a second apart (process start to exit, measured by the parent)",
                   "wall_s_of_the_three_runs": [round(w, 4) for w in walls]}

            def stats(text):
                for line in text.splitlines():
                    if line.startswith("[icsp_enc]"):
                        try:
                            return json.loads(line[len("[icsp_enc]"):])
                        except Exception:
                            return None
                return None
            e2e["stats"] = stats(out)
            e2e["stats_first_run_on_this_box"] = {k: v for k, v in (stats(first_out) or {}).items() if k in ("init_s", "hip_start_s", "setup_worker0", "map_files_s", "encode_s")}
            ref = next(s for s in golden if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", 300, 16, 0) and "bin_sha256" in s)
            binf = [f for f in os.listdir(tmp) if f.endswith(".bin")]
            e2e["bin_equals_reference"] = bool(binf) and hashlib.sha256(open(os.path.join(tmp, binf[0]), "rb").read()).hexdigest() == ref["bin_sha256"]
            ry = os.path.join(tmp, "test_yuv.yuv")
            e2e["recon_equals_reference"] = os.path.exists(ry) and hashlib.sha256(open(ry, "rb").read()).hexdigest() == ref["recon_sha256"]
            # steady state: the same clip ten times over (3000 frames, 456 MB in, 456 MB + bits out), --intraPeriod 10
            long_path = os.path.join(tmp, "long_cif(352X288)_3000f.yuv")
            with open(long_path, "wb") as fh:
                for _ in range(10):
                    fh.write(clip.tobytes())
            def long_run(period):
                """Two runs, the better one kept (both rates listed): whether one of the runtime's DMA engines has its ~6 ms
                first use inside the 20 ms after set-up varies from run to run."""
                best = None
                rates = []
                for _ in range(2):
                    time.sleep(1.0)
                    t0 = time.perf_counter()
                    r = run_enc(["-i", os.path.basename(long_path), "-n", "3000", "-q", "16", "--intraPeriod", str(period), "--stats"])
                    wall = time.perf_counter() - t0
                    st = stats(r.stdout.decode(errors="replace"))
                    rate = (st or {}).get("e2e_fps_excl_init", 0.0)
                    rates.append(rate)
                    if best is None or rate > best["stats"].get("e2e_fps_excl_init", 0.0):
                        best = {"rc": r.returncode, "wall_fps_incl_process_start_and_hip_init": round(3000 / wall, 1), "stats": st or {}}
                best["e2e_fps_excl_init_of_both_runs"] = rates
                return best
            e2e["long_3000f_ippp"] = long_run(10)
            e2e["long_3000f_all_intra"] = long_run(0)
            for f in (long_path, os.path.join(tmp, "long_compCIF_16_16_10.bin"), os.path.join(tmp, "long_compCIF_16_16_0.bin")):
                if os.path.exists(f):
                    os.remove(f)

    cpu = None
    if not (a.no_cpu or world > 1 or rank != 0):       # CPU baseline: rank 0 at N=1 only, on every core the process was given
        with all_cores():
            cpu = cpu_baseline()
            cpu["host_cores_available"] = len(_aff_all)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    traffic, pmc = _tj.get("k_intra_luma_bytes_per_launch"), _tj.get("k_intra_luma_sq")
    overlap = None
    try:
        import glob
        ov_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_overlap.json")))[-1]      # the latest round's kernel-trace summary
        ov = json.load(open(ov_path))
        ov["source"] = (ov.get("source") or "") + " [" + os.path.basename(ov_path) + "]"
        overlap = {k: ov[k] for k in ("launches_in_flight_median", "launch_duration_ms_median", "start_to_start_ms_median",
                                      "chip_level_GBps_from_trace", "source") if k in ov}
    except Exception:
        pass
    chip = achieved                                    # kernel's algorithmic bytes of a step / ms_per_step
    roof = {"bound": "hbm", "kernel": "k_intra_luma", "achieved": round(single_launch, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(single_launch / HBM_PEAK_GBS, 5), "traffic": traffic,
            "frac_is": "per launch (the contract's definition; unchanged since round 4).  Rounds 1-3 printed the chip-level figure under "
                       "this key: that one is chip_level.frac, kept beside it every round",
            "achieved_is": "the contract's per-launch figure: algorithmic_bytes_per_launch / avg_launch_ms, the launch's own duration "
                           "under HIP events on the stream it runs on (every 4th step of the timed region).  Two such launches are in "
                           "flight at any time (two batches on two streams, profiles/r*_overlap.json), each on about half the "
                           "chip's issue slots: see chip_level",
            "traffic_is": "HBM-side bytes of one launch of the kernel (rocprofv3 TCC counters of a 300-frame launch of the same kernel variant, "
                          "tools/pmc_workload.py -> profiles/traffic.json)",
            "traffic_over_algorithmic": round(traffic / (BYTES_INTRA_LUMA_KERNEL * NFRAMES), 3) if traffic else None,
            "launches_per_step": lps,
            "algorithmic_bytes_per_launch": BYTES_INTRA_LUMA_KERNEL * NFRAMES // lps, "avg_launch_ms": round(kern_ms, 4),
            "avg_launch_over": f"HIP events around the launches of every {EVENT_EVERY}th step of the timed regions ({n_ai} launches)",
            "chip_level": {"achieved": round(chip, 2), "frac": round(chip / HBM_PEAK_GBS, 5),
                           "is": "the kernel's algorithmic bytes of one step / ms_per_step (median region): what the chip as a whole "
                                 "moves for this kernel while launches of consecutive steps run side by side; avg_launch_ms > "
                                 "ms_per_step is that overlap", "kernel_trace": overlap},
            "whole_frame_read_frac": round(fps / world * BYTES_I_FRAME_READ / 1e9 / HBM_PEAK_GBS, 5),
            "whole_frame_rw_frac": round(fps / world * BYTES_I_FRAME_TOTAL / 1e9 / HBM_PEAK_GBS, 5),
            "limiter": "the contract's roofline is HBM; what binds this kernel is vector issue on a dependency chain: a CIF frame is 96 "
                       "wavefront steps (block rows chained in pairs; 114 in the plain form), a step is as long as its busiest SIMD needs to "
                       "issue the block tasks that landed on it (one task: about 720 vector + 220 scalar instructions; one wave issues an "
                       "instruction every 4 cycles), and with 2.3 frames per CU in flight the chip's issue slots are about 40 % full "
                       "(valu_issue_frac; 74-84 % on the loaded legs: config4.issue, config4.all_intra_loaded.issue, config5.issue) -- "
                       "DESIGN.md section 5"}
    if pmc and kern_ms > 0:
        # chip level, like `achieved`: the instructions of one step's launches (counters of one 300-frame launch of the same kernel
        # x launches per step) over the step's share of the timed region
        # executed fp64 vector instructions (rocprofv3 SQ counters, profiles/) x 64 lanes / time vs the un-fused peak
        ops = pmc.get("fp64_valu_insts_per_launch")
        if ops:
            roof["fp64_valu_frac"] = round(ops * lps * 64 / (step_ms * 1e-3) / 1e9 / FP64_VALU_PEAK_GOPS, 4)
            roof["fp64_valu_peak_Gops"] = round(FP64_VALU_PEAK_GOPS, 1)
            roof["fp64_valu_source"] = pmc.get("source")
        # every vector instruction of a wave holds its SIMD's issue port for >= 4 cycles (measured: 4.0-4.5; conversions to and
        # from f64 7.2; tools/probe_issue.hip), so wave-instructions / (SIMDs x clock / 4) is the share of the chip's vector
        # issue slots the luma kernel fills (the chroma kernels of the step are not in it) -- DESIGN.md section 5.
        vi = pmc.get("valu_insts_per_launch")
        if vi:
            peak = 256 * 4 * 2.4e9 / 4                        # 256 CUs x 4 SIMDs, one wave-instruction per 4 cycles at 2.4 GHz
            # conversions of one 300-frame launch of the luma kernel (counted), None when the counter file has none
            cvt_l = next((v.get("SQ_INSTS_VALU_CVT") for k, v in (_tj.get("kernels") or {}).items()
                          if k.startswith("k_intra_luma8") and v.get("frames_per_launch") == NFRAMES), None)
            roof["valu_issue_frac"] = round(issue_seconds(vi * lps, (ops or 0) * lps, cvt_l * lps if cvt_l else None, n_simd=4 * n_cu_dev) / (step_ms * 1e-3), 4)
            roof["valu_issue_frac_uniform_4_cycles"] = round(vi * lps / (step_ms * 1e-3) / peak, 4)
            roof["valu_insts_per_launch"] = int(vi)
            roof["waiting_share_of_wave_cycles"] = pmc.get("waiting_share_of_wave_cycles")
            roof["valu_issue_peak_Ginst"] = round(peak / 1e9, 1)
            # What binds, from the same counters: the chip issues at most `peak` vector wave-instructions a
            # second (1024 SIMDs, one per 4 cycles at 2.4 GHz).  A frame of this leg costs the luma kernel's instructions plus its
            # chroma kernels' (counted launches of k_chroma_dc and the strided k_residual8 of a 300-frame step, where the counter
            # file has them; else the luma kernel's x 1.5, their measured share in round 4); of those the fp64 multiplies / adds /
            # fused steps are what the arithmetic contract fixes -- everything else is overhead in principle.
            kk = _tj.get("kernels") or {}
            chroma_vi = chroma_f64 = None
            try:
                cd = next(v for k, v in kk.items() if k.startswith("k_chroma_dc@") and v.get("grid_threads") == 2 * NFRAMES * 256)
                cr = next(v for k, v in kk.items() if k.startswith("k_residual8_strided@") or k == "k_residual8@%d" % (256 * n_cu_dev))   # (one 256-thread workgroup per CU of THIS device)
                chroma_vi = cd["SQ_INSTS_VALU"] + cr["SQ_INSTS_VALU"]
                chroma_f64 = cd.get("fp64_valu_insts_per_launch", 0) + cr.get("fp64_valu_insts_per_launch", 0)
            except Exception:
                pass
            insts_frame = (vi + chroma_vi) / NFRAMES if chroma_vi else vi * 1.5 / NFRAMES
            f64_frame = ((ops or 0) + chroma_f64) / NFRAMES if chroma_f64 else (ops or 0) * 1.5 / NFRAMES
            fps_gpu = fps / world
            nsimd = 4 * n_cu_dev
            cvt_frame = (cvt_l / NFRAMES * (insts_frame * NFRAMES / vi)) if cvt_l else None       # (the luma kernel's share, scaled to the step's kernels)
            t_frame = issue_seconds(insts_frame, f64_frame, cvt_frame, nsimd)                     # seconds of chip issue time per frame, as the kernels' mixes run
            t_best = issue_seconds(insts_frame, f64_frame, cvt_frame, nsimd, best_case=True)      # ... if every simple integer instruction issued at its 2-cycle rate
            t_floor = issue_seconds(f64_frame, f64_frame, 0.0, nsimd)
            roof["binding"] = {
                "resource": "valu_issue", "frac": round(fps_gpu * t_frame, 4),
                "valu_insts_per_frame": int(insts_frame), "fp64_floor_insts_per_frame": int(f64_frame),
                "ceiling_fps_at_current_insts": round(1.0 / t_frame, 1), "ceiling_fps_at_current_insts_best_case": round(1.0 / t_best, 1),
                "ceiling_fps_at_fp64_floor": round(1.0 / t_floor, 1) if f64_frame else None,
                "ceiling_fps_uniform_4_cycles": round(peak / insts_frame, 1),
                "value_over_ceiling_at_current_insts": round(fps_gpu * t_frame, 4),
                "value_over_ceiling_at_fp64_floor": round(fps_gpu * t_floor, 4) if f64_frame else None,
                "issue_ns_per_class": issue_costs()["ns"],
                "chroma_kernels": "counted" if chroma_vi else "luma kernel x 1.5 (no counted chroma launches in profiles/traffic.json)",
                "is": "vector wave-instructions per frame of this leg (luma kernel + the step's chroma kernels, rocprofv3 SQ counters in "
                      "profiles/traffic.json) priced per class at what a SIMD's issue port was MEASURED to sustain with 2-4 waves on it "
                      "(tools/probe_issue_waves.hip, profiles/r06_probe_issue_waves.txt: fp64 / conversions / quarter-rate 32-bit about 4.3 "
                      "cycles, simple integer 2.3 cycles alone and about 3.7 between fp64 instructions): `frac` of the chip's issue time is "
                      "filled; at today's instruction count the chip could do ceiling_fps_at_current_insts frames/s (…_best_case: every simple "
                      "integer instruction at the 2.3 cycles that only a stream of ONE opcode reaches -- no mixed stream does; "
                      "…_uniform_4_cycles: rounds 4-5's model), with nothing but "
                      "the bit-exact fp64 arithmetic left ceiling_fps_at_fp64_floor.  The HBM figures above stay the contract's; this is "
                      "the ceiling that binds (DESIGN.md section 5)"}
    line = {
        "metric": "CIF encode fps, resident encode loop (all-intra QP=16; IPPP and the 8-GPU workloads alongside)", "value": round(fps, 1),
        "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4),
        "value_is": "median of the repeats of the K-step timed region (each bracketed by barrier + synchronize)",
        "repeats": spread(dts_ai, world * NFRAMES * a.steps),
        "value_three_batches": round(world * NFRAMES * a.steps / med(dts_r3), 1),
        "three_batches": {"is": "the same K-step regions with THREE resident 300-frame batches in rotation instead of two (a ring of three "
                                "chunk slots): three batches in flight on three chain streams; secondary, `value` keeps its two-batch regime",
                          "ms_per_step": round(med(dts_r3) / a.steps * 1e3, 4), "repeats": spread(dts_r3, world * NFRAMES * a.steps),
                          "regime": choice_r3},
        "value_same_range": round(world * NFRAMES * a.steps / med(dts_same), 1),
        "same_range": {"is": "the regime of rounds 1-2, for like-for-like comparison across rounds: ONE resident 300-frame batch "
                             "encoded again and again (two parts on two streams), K steps, same repeats",
                       "ms_per_step": round(med(dts_same) / a.steps * 1e3, 4), "repeats": spread(dts_same, world * NFRAMES * a.steps),
                       "regime": choice_same,
                       "roofline_per_launch": {"avg_launch_ms": round(ms_same / max(n_same, 1), 4), "launches": n_same,
                                               "achieved": round(BYTES_INTRA_LUMA_KERNEL * NFRAMES / max(1, round(n_same / max(1, n_ev_steps))) / (ms_same / max(n_same, 1) * 1e-3) / 1e9, 2) if n_same else None}},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "foremanlike_cif 352x288 300f, --intraPeriod 0 (all-intra), QP=16, per GPU (BASELINE configs[1]); steps "
                               "alternate between two disjoint resident 300-frame batches (frames 0-299 and 300-599 of the sequence)",
                   "frames_per_step_per_gpu": NFRAMES, "parallelism": f"frame/GOP shards over {world} GPU(s), no collectives",
                   "timed_region": "resident transform/prediction kernels of the encode loop; entropy packing, PCIe and file I/O are "
                                   "reported separately (device_pack, e2e) and are part of cpu_baseline's whole-process figure"},
        "roofline": roof,
        "cpu_baseline": cpu,
        # the oracle's C restatement on every core this process was given (a library call: no file I/O, no bitstream writing) -- the
        # same arithmetic as the reference binary above, without its four-thread limit
        "cpu_baseline_port": (cpu or {}).get("port_all_cores") if (cpu or {}).get("kind") == "reference" else (cpu if cpu else None),
        "parity": parity,
        "kernels_ms_per_step": {k: round(v[0], 4) for k, v in prof.items() if v[1]},
        "regime": dict(choice_ai, frames_in_flight_per_cu=round(2 * NFRAMES / 256, 2), rank0_gpu_numa_node=_node, rank0_host_thread_bound=bool(_bound.value),
                       note="two independent 300-frame batches in flight on two streams; no scaling curve over GPUs has been measured "
                            "on hardware so far (the driver's multi-GPU node has not been available: SCALE_r01..r04 are skipped records)"),
        "isolated_pass": {"ms": round(iso_ai, 4), "fps_per_gpu": round(NFRAMES / iso_ai * 1e3, 1),
                          "note": "one pass at a time, the host waiting before and after each (includes a launch and a sync round trip); in "
                                  "the timed steps consecutive passes over independent batches run side by side (DESIGN.md section 4)"},
        "psnr_y_db": round(psnr_ai, 4),
        "pcie_inclusive_fps": round(NFRAMES / pcie_dt, 1),
        "pcie_inclusive_fps_is": "the one-call boundary with the caller's arrays in pinned memory (pcie_inclusive.pinned_fps; from plain memory: "
                                 "pageable_fps, which is what this key held until round 3 -- one call, plain memory) -- never `value`",
        "pcie_inclusive_fps_pageable": pcie.get("pageable_fps"),
        "pcie_inclusive": pcie,
        "small_ranges": small,
        "device_pack": {"bin_bytes": 14 + nbits // 8 + 1, "kernels_ms": round(pack_ms[0] / max(pack_ms[1], 1), 4),
                        "pack_and_copy_ms": round(pack_dt * 1e3, 3), "upload_encode_pack_fps": round(NFRAMES / e2e_dt, 1),
                        "note": "5 kernels (count, 2 scans, zero, pack) + D2H of the bits only; pcie_inclusive_fps copies "
                                "levels/flags/vectors/recon back instead"},
        "decode": {"all_intra_fps": round(dec_ai_fps, 1),
                   "note": "device reconstruction of the resident syntax of the same 300 frames (icsp_decode_resident), this rank"},
        "ippp": ippp, "config4": config4, "config5": config5, "e2e": e2e,
    }
    emit(line)


if __name__ == "__main__":
    main()
