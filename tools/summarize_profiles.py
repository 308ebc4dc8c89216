#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh (merged under gpurun_out/<tag>/) into the tracked summaries
under profiles/:

    python tools/summarize_profiles.py gpurun_out/<tag> <tag>

  profiles/<tag>_kernel_stats_bench.csv   rocprofv3 --kernel-trace --stats of the bench command (per-kernel calls / average ns)
  profiles/<tag>_bench_profiled.json      the bench line printed under the profiler
  profiles/traffic.json                   per kernel launch: HBM-side bytes from the TCC counters, the calibration of those
                                          counters on known byte counts, SQ instruction / cycle counters, derived fractions

Traffic: rocprofv3's FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM), i.e. reports half the bytes of a wide coalesced read; the calibration run (tools/calib_traffic.hip,
512 MiB moved once by 2-, 4-, 8-, 16-byte-per-lane accesses and by the codec's 8-byte block rows) gives the factor for each
access width, and the request-size counters TCC_EA0_RDREQ_{32B,64B,128B} give the bytes without any factor:
    read bytes  = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B        (checked against the known byte counts below)
    write bytes = WRITE_SIZE * 1024
"""
import collections
import csv
import re
import glob
import json
import os
import shutil
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, NMB = 352 * 288, 396
KERNELS = ("k_dec_intra_luma32", "k_dec_serial", "k_dec_blocks", "k_intra_luma32", "k_intra_luma8", "k_chroma_dc", "k_residual8", "k_me<false",
           "k_me<true", "k_serial_fused", "k_frame_serial", "k_bits_count", "k_bits_scan", "k_chunk_base", "k_pack_zero", "k_pack",
           "cal_read_rows8", "cal_read", "cal_write")


def short(n):
    m = re.search(r"k_intra_luma8<(\d+), (true|false), (\d+)(?:, (?:true|false))?>", n)      # (round 5: a fourth argument, the quantiser form)
    if m:                                                   # the 8-lane luma kernel by variant: waves per workgroup, rows chained in groups of
        return "k_intra_luma8" + (f"_w{m.group(1)}g{m.group(3)}" if m.group(3) != "0" else "")
    for k in KERNELS:
        if k in n:
            s = k.replace("<", "_").rstrip("_")
            if s in ("cal_read", "cal_write"):
                for t, b in (("unsigned int, 2", 8), ("unsigned int, 4", 16), ("unsigned short", 2), ("unsigned int", 4)):
                    if t in n:
                        s += f"_{b}B"
                        break
            return s
    return n.split("(")[0][-40:]


def counters(sub):
    """{(kernel, grid): {counter: mean value over dispatches}} of one PMC pass directory."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Kernel_Name"]:
                continue
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"_launches": max(len(v) for v in d.values())} for k, d in agg.items()}


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench.csv"))
bj = os.path.join(out_dir, "bench_profiled.json")
if os.path.exists(bj) and os.path.getsize(bj):
    shutil.copy(bj, os.path.join(ROOT, "profiles", f"{tag}_bench_profiled.json"))

# ---- calibration on known byte counts
cal = {}
KNOWN = 512 * 1024 * 1024
for sub in sorted(glob.glob(os.path.join(out_dir, "cal_*"))):
    if not os.path.isdir(sub):
        continue
    for (kn, grid), d in counters(os.path.basename(sub)).items():
        e = cal.setdefault(kn, {"known_bytes": 5000 * P if "rows8" in kn else KNOWN})
        for c, v in d.items():
            if not c.startswith("_"):
                e[c] = v
for kn, e in cal.items():
    kb = e["known_bytes"]
    if "FETCH_SIZE" in e and "read" in kn:
        e["bytes_per_FETCH_SIZE_KiB_unit"] = round(kb / e["FETCH_SIZE"], 1)          # 1024 would be exact; 2048 = "reports half"
    if "WRITE_SIZE" in e and "write" in kn:
        e["bytes_per_WRITE_SIZE_KiB_unit"] = round(kb / e["WRITE_SIZE"], 1)
    if "TCC_EA0_RDREQ_32B" in e and "read" in kn:
        rd = 32 * e.get("TCC_EA0_RDREQ_32B", 0) + 64 * e.get("TCC_EA0_RDREQ_64B", 0) + 128 * e.get("TCC_EA0_RDREQ_128B", 0)
        e["rdreq_sized_bytes_over_known"] = round(rd / kb, 4)
    if "TCC_EA0_WRREQ" in e and "write" in kn:
        e["wrreq_x64_over_known"] = round(64 * e["TCC_EA0_WRREQ"] / kb, 4)

# ---- the codec's kernels
a, b, c, d = counters("pmc_a"), counters("pmc_b"), counters("pmc_c"), counters("pmc_d")
keys = sorted(set(a) | set(b) | set(c) | set(d))
# read factor for FETCH_SIZE in the codec's access shape (8-byte block rows), from the calibration; 2048 if it was not run
f_rows = cal.get("cal_read_rows8", {}).get("bytes_per_FETCH_SIZE_KiB_unit", 2048.0)
kernels = {}
for k in keys:
    kn, grid = k
    e = {"grid_threads": grid, "launches_seen": int(max(x.get(k, {}).get("_launches", 0) for x in (a, b, c, d)))}
    for src in (a, b, c, d):
        for cn, v in src.get(k, {}).items():
            if not cn.startswith("_"):
                e[cn] = round(v, 1)
    if "TCC_EA0_RDREQ_32B" in e:
        e["read_bytes_per_launch"] = int(32 * e.get("TCC_EA0_RDREQ_32B", 0) + 64 * e.get("TCC_EA0_RDREQ_64B", 0) + 128 * e.get("TCC_EA0_RDREQ_128B", 0))
    elif "FETCH_SIZE" in e:
        e["read_bytes_per_launch"] = int(e["FETCH_SIZE"] * f_rows)
    if "WRITE_SIZE" in e:
        e["write_bytes_per_launch"] = int(e["WRITE_SIZE"] * 1024)
    if "read_bytes_per_launch" in e and "write_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
    f64 = e.get("SQ_INSTS_VALU_ADD_F64", 0) + e.get("SQ_INSTS_VALU_MUL_F64", 0) + e.get("SQ_INSTS_VALU_FMA_F64", 0)
    if "SQ_INSTS_VALU" in e:
        e["fp64_valu_insts_per_launch"] = int(f64)
        e["fp64_share_of_valu_insts"] = round(f64 / max(e["SQ_INSTS_VALU"], 1), 4)
    if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_INST_ANY" in e:
        e["issue_stall_share_of_wave_cycles"] = round(e["SQ_WAIT_INST_ANY"] / max(e["SQ_WAVE_CYCLES"], 1), 4)
        e["waiting_share_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0) / max(e["SQ_WAVE_CYCLES"], 1), 4)
        e["valu_active_share_of_wave_cycles"] = round(e.get("SQ_ACTIVE_INST_VALU", 0) / max(e["SQ_WAVE_CYCLES"], 1), 4)
    kernels[f"{kn}@{grid}"] = e

# algorithmic bytes of the launches the bench line and VERDICT quote (300 CIF frames; 15 frames per P-step launch)
ALG = {"k_intra_luma32": lambda fr: fr * (4 * P + 8 * NMB), "k_intra_luma8": lambda fr: fr * (4 * P + 8 * NMB), "k_intra_luma8_w4g2": lambda fr: fr * (4 * P + 8 * NMB), "k_me_false": lambda fr: fr * (3 * P + 64 * NMB),
       "k_residual8": lambda fr: fr * (3 * P + P * 3 // 2 + 3 * P + 8 * NMB)}
for name, e in kernels.items():
    kn, grid = name.split("@")
    grid = int(grid)
    frames = None
    if kn == "k_intra_luma32":
        frames = grid // 512 if grid >= 512 * 256 else grid // 704          # <8,4> above 256 frames, else <11,1>
    elif kn == "k_intra_luma8" and grid % 192 == 0:
        frames = grid // 192                                                 # CIF: three waves per frame
    elif kn == "k_intra_luma8_w4g2" and grid % 256 == 0:
        frames = grid // 256                                                 # CIF, rows in pairs: four waves per frame
    elif kn == "k_me_false":
        frames = 15                                                          # configs[2]: 30 GOPs in two groups (the grid is padded to 16 frames)
    elif kn == "k_residual8" and 900 < grid // 256 < 1400:
        frames = 15                                                          # P-step launch of configs[2]
    if frames and kn in ALG and "hbm_bytes_per_launch" in e:
        e["frames_per_launch"] = frames
        e["algorithmic_bytes_per_launch"] = ALG[kn](frames)
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 3)

out = {"method": __doc__.split("Traffic:")[1].strip(), "calibration": cal, "kernels": kernels}
# The dominant kernel of the bench line: the 300-frame launch of k_intra_luma8 (three waves per CIF frame: 57600 threads) that the
# alternating-batches regime makes; round 2's k_intra_luma32<8,4> launch (153600 threads) is kept beside it.
# The SQ counters of a dispatch may cover only part of its waves (SQ_WAVES says how many): instruction counts are scaled to the whole
# launch by expected waves / SQ_WAVES.
def sq_entry(bk, waves_expected):
    e = kernels[bk]
    cov = (e.get("SQ_WAVES", 0) / waves_expected) if waves_expected and e.get("SQ_WAVES") else 1.0
    sc = 1.0 / cov if cov > 0 else 1.0
    return {"fp64_valu_insts_per_launch": int(e.get("fp64_valu_insts_per_launch", 0) * sc), "valu_insts_per_launch": int(e.get("SQ_INSTS_VALU", 0) * sc),
            "sq_wave_coverage": round(cov, 3), "issue_stall_share_of_wave_cycles": e.get("issue_stall_share_of_wave_cycles"),
            "waiting_share_of_wave_cycles": e.get("waiting_share_of_wave_cycles"),
            "source": f"rocprofv3 --pmc SQ_INSTS_VALU_{{ADD,MUL,FMA}}_F64 on {bk}, scaled by expected waves / SQ_WAVES (profiles/traffic.json, tools/profile_round.sh)"}
# (round 4: the headline's launch is the pairs variant, four waves per frame; a build that picks the plain wavefront has 57600 threads)
dom, dom_waves = "k_intra_luma8_w4g2@76800", 300 * 4
if dom not in kernels:
    dom, dom_waves = "k_intra_luma8@57600", 300 * 3
old = "k_intra_luma32@153600"
if dom in kernels and "hbm_bytes_per_launch" in kernels[dom]:
    out["k_intra_luma_kernel"] = dom
    out["k_intra_luma_bytes_per_launch"] = kernels[dom]["hbm_bytes_per_launch"]
    out["k_intra_luma_algorithmic_bytes_per_launch"] = 300 * (4 * P + 8 * NMB)
    out["k_intra_luma_sq"] = sq_entry(dom, dom_waves)
if old in kernels and "hbm_bytes_per_launch" in kernels[old]:
    out["k_intra_luma32_bytes_per_launch"] = kernels[old]["hbm_bytes_per_launch"]
    out["k_intra_luma32_sq"] = sq_entry(old, 300 * 8)
    if "k_intra_luma_sq" not in out:
        out["k_intra_luma_kernel"] = old
        out["k_intra_luma_bytes_per_launch"] = kernels[old]["hbm_bytes_per_launch"]
        out["k_intra_luma_algorithmic_bytes_per_launch"] = 300 * (4 * P + 8 * NMB)
        out["k_intra_luma_sq"] = out["k_intra_luma32_sq"]
# ---- the loaded legs of the bench line (tools/leg_workload.py under rocprofv3): per PASS sums over every launch of the codec's kernels
def leg_counters(sub):
    """{kernel: {counter: SUM over dispatches}} and dispatch counts of one PMC pass directory of a leg"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Kernel_Name"]:
                continue
            kn = short(r["Kernel_Name"])
            agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[kn][r["Counter_Name"]] += 1
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":          # the dispatch's own duration, for the clock it ran at
                agg[kn]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, {k: max(v.values()) for k, v in nd.items()}


legs = {}
for leg, (lp, lnmb, per) in {"config4": (P, NMB, 10), "config4_allintra": (P, NMB, 1), "config5": (1920 * 1088, 8160, 30)}.items():
    info = None
    ip = os.path.join(out_dir, f"leg_{leg}.json")
    if os.path.exists(ip):
        for l in open(ip):
            if l.startswith("{"):
                info = json.loads(l)
    tot, by = collections.defaultdict(float), {}
    seen = False
    for sub in ("a", "b", "c"):
        agg, nd = leg_counters(f"leg_{leg}_pmc_{sub}")
        for kn, cs in agg.items():
            seen = True
            e = by.setdefault(kn, {})
            e["launches"] = max(e.get("launches", 0), nd[kn])
            for cn, v in cs.items():
                e[cn] = v
                tot[cn] += v
    if not seen or not info:
        continue
    passes = info["passes_total"]
    frames = info["frames"]
    rd = 32 * tot["TCC_EA0_RDREQ_32B"] + 64 * tot["TCC_EA0_RDREQ_64B"] + 128 * tot["TCC_EA0_RDREQ_128B"]
    wr = tot["WRITE_SIZE"] * 1024
    f64 = tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + tot["SQ_INSTS_VALU_FMA_F64"]
    n_i = frames if per == 1 else (frames + per - 1) // per
    n_p = frames - n_i
    alg = n_i * (6 * lp + 10 * lnmb) + n_p * ((3 * lp + 64 * lnmb) + 84 * lnmb + (7 * lp + lp // 2 + 8 * lnmb))
    legs[leg] = {"frames": frames, "passes_counted": passes,
                 "valu_insts_per_pass": int(tot["SQ_INSTS_VALU"] / passes), "fp64_valu_insts_per_pass": int(f64 / passes),
                 "waiting_share_of_wave_cycles": round(tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1), 4),
                 "valu_active_share_of_wave_cycles": round(tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1), 4),
                 "hbm_bytes_per_pass": int((rd + wr) / passes), "read_bytes_per_pass": int(rd / passes), "write_bytes_per_pass": int(wr / passes),
                 "algorithmic_bytes_per_pass": int(alg), "traffic_over_algorithmic": round((rd + wr) / passes / alg, 3) if rd and wr else None,
                 "ms_per_pass_of_the_unprofiled_run": info.get("ms_per_pass_here"), "choice": info.get("choice"),
                 # GRBM_GUI_ACTIVE sums the 8 XCDs (MI355X_MICROARCH.md, DVFS): cycles / 8 / the dispatches' own time; long dispatches only
                 "effective_clock_GHz_under_pmc": round(sum(e.get("GRBM_GUI_ACTIVE", 0) for e in by.values() if e.get("_ns", 0) / max(e.get("launches", 1), 1) > 3e5) / 8 /
                                                        max(sum(e["_ns"] for e in by.values() if e.get("_ns", 0) / max(e.get("launches", 1), 1) > 3e5), 1), 3),
                 "by_kernel": {kn: {"launches_per_pass": round(e["launches"] / passes, 1), "valu_insts_per_pass": int(e.get("SQ_INSTS_VALU", 0) / passes),
                                    "fp64_share": round((e.get("SQ_INSTS_VALU_ADD_F64", 0) + e.get("SQ_INSTS_VALU_MUL_F64", 0) + e.get("SQ_INSTS_VALU_FMA_F64", 0)) / max(e.get("SQ_INSTS_VALU", 1), 1), 3),
                                    "waiting_share": round(e.get("SQ_WAIT_ANY", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                    "hbm_bytes_per_pass": int((32 * e.get("TCC_EA0_RDREQ_32B", 0) + 64 * e.get("TCC_EA0_RDREQ_64B", 0) + 128 * e.get("TCC_EA0_RDREQ_128B", 0) + 1024 * e.get("WRITE_SIZE", 0)) / passes)}
                               for kn, e in sorted(by.items()) if e.get("SQ_INSTS_VALU")},
                 "source": f"rocprofv3 --pmc (three passes of their own) on tools/leg_workload.py {leg}: sums over every dispatch / {passes} passes (tools/profile_round.sh)"}
    st = glob.glob(os.path.join(out_dir, f"leg_{leg}_stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{leg}.csv"))
if legs:
    out["legs"] = legs
else:
    try:        # a run without the leg passes keeps what an earlier run of the round measured
        prev = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if "legs" in prev:
            out["legs"] = prev["legs"]
    except Exception:
        pass
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for leg, e in legs.items():
    print("leg", leg, {k: v for k, v in e.items() if k not in ("by_kernel", "source", "choice")})
    for kn, ke in e["by_kernel"].items():
        print("   ", kn, ke)
print(json.dumps({"calibration": cal}, indent=1))
for k, e in kernels.items():
    print(k, {x: e[x] for x in ("launches_seen", "read_bytes_per_launch", "write_bytes_per_launch", "traffic_over_algorithmic", "fp64_share_of_valu_insts",
                                "issue_stall_share_of_wave_cycles", "valu_active_share_of_wave_cycles") if x in e})
