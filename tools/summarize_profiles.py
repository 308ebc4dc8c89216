#!/usr/bin/env python3
"""Condense rocprofv3 outputs merged under gpurun_out/ into the tracked summaries under profiles/.

    python tools/summarize_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_json> <tag>

Traffic: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, the gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE
reports half the bytes of a coalesced read; confirmed here on __amd_rocclr_copyBuffer)."""
import collections, csv, glob, json, os, shutil, sys

stats_dir, fdir, wdir, bench_json, tag = sys.argv[1:6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, NMB = 352 * 288, 396

def short(n):
    for k in ("k_dec_intra_luma32", "k_dec_serial", "k_dec_blocks", "k_intra_luma32", "k_chroma_dc", "k_residual8", "k_me<false",
              "k_me<true", "k_serial_fused", "k_frame_serial", "k_bits_count", "k_bits_scan", "k_chunk_base", "k_pack_zero", "k_pack"):
        if k in n:
            return k.replace("<", "_").rstrip("_")
    return n.split("(")[0][-40:]

shutil.copy(glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench.csv"))
shutil.copy(bench_json, os.path.join(ROOT, "profiles", f"{tag}_bench.json"))
rows = collections.defaultdict(dict)
copy_cal = {}
for name, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0])):
        if r["Counter_Name"] != name:
            continue
        if "copyBuffer" in r["Kernel_Name"] and r["Grid_Size"] == "65536":
            copy_cal.setdefault(name, []).append(float(r["Counter_Value"]))
        if "rocclr" in r["Kernel_Name"]:
            continue
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        rows[k][name] = sum(v) / len(v)
        rows[k]["launches"] = len(v)
alg = {"k_intra_luma32": lambda frames: frames * (4 * P + 8 * NMB)}
out = {"method": __doc__.split("Traffic:")[1].strip(), "unit_of_counters": "KB",
       "copy_calibration_1MiB": {k: sum(v) / len(v) for k, v in copy_cal.items()}, "kernels": {}}
for (kn, grid), d in sorted(rows.items()):
    f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    e = {"grid_threads": grid, "launches_seen": d["launches"], "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    out["kernels"][f"{kn}@{grid}"] = e
# the bench's dominant kernel: the 300-frame all-intra launch of k_intra_luma32 (largest grid)
big = max((k for k in out["kernels"] if k.startswith("k_intra_luma32")), key=lambda k: out["kernels"][k]["grid_threads"])
out["k_intra_luma_bytes_per_launch"] = out["kernels"][big]["hbm_bytes_per_launch"]
out["k_intra_luma_algorithmic_bytes_per_launch"] = 300 * (4 * P + 8 * NMB)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(big, out["kernels"][big], out["copy_calibration_1MiB"])
