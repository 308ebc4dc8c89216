#!/usr/bin/env python3
"""Merge the per-geometry outputs of tools/sweep_regimes.py into one tracked table: merge_sweeps.py out.json in1.json in2.json ..."""
import json, sys
out, ins = sys.argv[1], sys.argv[2:]
rows, wall, budget = [], 0.0, None
for p in ins:
    d = json.load(open(p))
    for r in d["rows"]:                                    # a later file's measurement of a regime replaces an earlier one
        key = (r["geometry"], r["batch_cif_equivalent"], r["period"], r["ranges"])
        rows = [x for x in rows if (x["geometry"], x["batch_cif_equivalent"], x["period"], x["ranges"]) != key] + [r]
    wall += d.get("wall_s", 0); budget = d.get("budget_s_per_measurement", budget)
below = [r for r in rows if r["default_over_best"] < 0.97]
by_geo = {}
for r in rows:
    g = by_geo.setdefault(r["geometry"], {"regimes": 0, "worst_default_over_best": 1.0, "below_0.97": 0})
    g["regimes"] += 1; g["worst_default_over_best"] = min(g["worst_default_over_best"], r["default_over_best"]); g["below_0.97"] += r["default_over_best"] < 0.97
res = {"tool": "tools/sweep_regimes.py (one run per geometry group, merged by tools/merge_sweeps.py)", "budget_s_per_measurement": budget,
       "regimes": len(rows), "wall_s": round(wall, 1), "worst_default_over_best": min(r["default_over_best"] for r in rows),
       "by_geometry": by_geo,
       "regimes_below_0.97": [{k: r[k] for k in ("geometry", "batch_cif_equivalent", "frames_per_range", "period", "ranges", "default_fps", "best_forced", "default_over_best")} for r in below],
       "rows": rows}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "rows"}, indent=1))
