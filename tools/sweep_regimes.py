#!/usr/bin/env python3
"""Is the library's default scheduling within a few percent of the best forced setting in every regime?  

    sweep_regimes.py [--quick] [--out profiles/r05_sweep.json] [--budget-s 0.12]

Regimes: geometry {CIF, 352x576, 4CIF, 720p, 1088p} x batch {100 ... 3390 CIF frames' worth of macroblocks} x {all-intra, period
10} x {one resident range encoded again and again, two alternating, three in rotation}.  For each regime the resident throughput
with everything left to the library (default) and with one knob forced at a time:

    all-intra: ICSP_INTRA_FORM 8 (plain wavefront) / 32, ICSP_INTRA_GROUP 2 / 1 (rows chained in pairs / the plain wavefront), ICSP_CHROMA_CAP 0, ICSP_WHOLE 0 (two ranges or more), ICSP_I_GROUPS 1 (one range), ICSP_CHAINS3 0 (three ranges)
    period 10: ICSP_P_GROUPS 1 / 2, ICSP_WHOLE 0 and ICSP_I_STREAM_B 0 (two ranges or more), ICSP_INTRA_FORM 8 / 32, ICSP_INTRA_GROUP 2 (the I step)

Every setting produces the same bytes (tests/); this is about speed only.  Writes the table and a summary (worst default / best
ratio, the regimes below 0.97) as JSON.  --quick: ten regimes -- the headline ones and the closest calls of profiles/r05_sweep.json --
each against the knobs that came closest in profiles/r05_sweep.json, the default measured before AND after the forced settings (the -m gpu smoke test)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from icspcodec_amd import capi, clipgen  # noqa: E402

GEOMS = {"CIF": (352, 288), "352x576": (352, 576), "4CIF": (704, 576), "720p": (1280, 720), "1088p": (1920, 1088)}
BATCHES = [100, 200, 250, 270, 300, 350, 400, 600, 1000, 3390]          # CIF frames' worth of macroblocks
KNOBS_AI = [("ICSP_INTRA_FORM", "8"), ("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_GROUP", "2"), ("ICSP_INTRA_GROUP", "1"), ("ICSP_CHROMA_CAP", "0"), ("ICSP_WHOLE", "0"), ("ICSP_I_GROUPS", "1"), ("ICSP_CHAINS3", "0")]
KNOBS_IP = [("ICSP_P_GROUPS", "1"), ("ICSP_P_GROUPS", "2"), ("ICSP_WHOLE", "0"), ("ICSP_INTRA_FORM", "8"), ("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_GROUP", "2"), ("ICSP_I_STREAM_B", "0")]
# --quick: (geometry, batch, period, ranges, knobs to force); the headline regimes and profiles/r05_sweep.json's lowest default / best ratios
QUICK = [("CIF", 300, 0, 2, [("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_GROUP", "1"), ("ICSP_CHROMA_CAP", "0")]),
         ("CIF", 300, 10, 2, [("ICSP_WHOLE", "0"), ("ICSP_INTRA_FORM", "8"), ("ICSP_I_STREAM_B", "0")]),
         ("CIF", 3390, 0, 1, [("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_GROUP", "1")]),
         ("CIF", 300, 0, 3, [("ICSP_CHAINS3", "0"), ("ICSP_CHROMA_CAP", "0")]),
         ("4CIF", 350, 10, 3, [("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_FORM", "8")]),
         ("CIF", 250, 10, 3, [("ICSP_INTRA_GROUP", "2"), ("ICSP_INTRA_FORM", "32")]),
         ("352x576", 400, 0, 3, [("ICSP_CHROMA_CAP", "0"), ("ICSP_INTRA_GROUP", "2")]),
         ("720p", 270, 0, 3, [("ICSP_INTRA_GROUP", "1"), ("ICSP_CHAINS3", "0")]),
         ("CIF", 1000, 10, 3, [("ICSP_INTRA_FORM", "32"), ("ICSP_P_GROUPS", "1")]),
         ("4CIF", 1000, 10, 1, [("ICSP_INTRA_FORM", "32"), ("ICSP_INTRA_FORM", "8")])]
_clips = {}


def base_clip(w, h, period):
    key = (w, h, bool(period))
    if key not in _clips:
        name = "stefanlike" if period else "foremanlike"
        _clips[key] = clipgen.synth_clip(name, 10, width=w, height=h) if (w, h) != (352, 288) else clipgen.synth_clip(name, 30)
    return _clips[key]


def measure(w, h, qp, period, n, ranges, env, budget_s):
    old = {k: os.environ.get(k) for k, _ in env}
    for k, v in env:
        os.environ[k] = v
    try:
        enc = capi.Encoder(w, h, qp, qp, period, max_frames=n * ranges)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        base = base_clip(w, h, period)
        for f in range(0, n * ranges, len(base)):
            enc.upload(base[: min(len(base), n * ranges - f)], first=f)
        k = 0
        t0 = time.perf_counter()
        while k < 4 or time.perf_counter() - t0 < 0.04:                # settle: clocks, first launches, stream creation
            enc.encode_resident((k % ranges) * n, n)
            k += 1
        enc.sync()
        per = (time.perf_counter() - t0) / k
        passes = max(6, min(400, int(budget_s / max(per, 1e-5))))
        passes += (-passes) % ranges
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            for j in range(passes):
                enc.encode_resident((j % ranges) * n, n)
            enc.sync()
            best = max(best, n * passes / (time.perf_counter() - t0))
        return best, enc.last_choice()
    finally:
        enc.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--budget-s", type=float, default=0.12)
    ap.add_argument("--geoms", default=",".join(GEOMS))
    ap.add_argument("--regimes", default="", help="only these: geometry:batch:period:ranges,...")
    ap.add_argument("--periods", default="0,10", help="of the full sweep: only these intra periods")
    ap.add_argument("--ranges", default="1,2,3", help="of the full sweep: only these numbers of ranges in rotation")
    ap.add_argument("--batches", default=",".join(str(b) for b in BATCHES), help="of the full sweep: only these batch sizes")
    a = ap.parse_args()
    regimes = []
    quick_knobs = {}
    if a.regimes:
        for x in a.regimes.split(","):
            g, b, per, r = x.split(":")
            regimes.append((g, int(b), int(per), int(r)))
    elif a.quick:
        regimes = [q[:4] for q in QUICK]
        quick_knobs = {q[:4]: q[4] for q in QUICK}
    else:
        for g in a.geoms.split(","):
            for b in [int(x) for x in a.batches.split(",")]:
                for period in [int(x) for x in a.periods.split(",")]:
                    for r in [int(x) for x in a.ranges.split(",")]:
                        regimes.append((g, b, period, r))
    rows = []
    t_all = time.time()
    for g, b, period, r in regimes:
        w, h = GEOMS[g]
        nmb = (w // 16) * (h // 16)
        n = max(1, round(b * 396 / nmb))
        if period:
            n = max(period, n // period * period)
        if n * r * (w * h * 13) > 60e9:                                 # device memory of the resident ranges
            continue
        qp = 8 if period else 16
        knobs = quick_knobs.get((g, b, period, r)) or [kv for kv in (KNOBS_IP if period else KNOBS_AI)
                                                        if not (kv[0] in ("ICSP_WHOLE", "ICSP_I_STREAM_B") and r == 1) and not (kv[0] == "ICSP_I_GROUPS" and r > 1) and not (kv[0] == "ICSP_CHAINS3" and r < 3)]
        try:
            dflt, choice = measure(w, h, qp, period, n, r, [], a.budget_s)
            forced = {}
            for kv in knobs:
                forced["%s=%s" % kv], _ = measure(w, h, qp, period, n, r, [kv], a.budget_s)
            # (the default once more behind the forced settings: as many chances as a pair of them has -- ADVICE r04)
            dflt = max(dflt, measure(w, h, qp, period, n, r, [], a.budget_s)[0])
        except Exception as e:                                          # (a geometry / batch the box cannot hold)
            print("skip", g, b, period, r, e, flush=True)
            continue
        best_k = max(forced, key=forced.get)
        best = max(dflt, forced[best_k])
        row = {"geometry": g, "batch_cif_equivalent": b, "frames_per_range": n, "period": period, "ranges": r, "default_fps": round(dflt, 1),
               "default_choice": choice, "forced_fps": {k: round(v, 1) for k, v in forced.items()}, "best_forced": best_k,
               "default_over_best": round(dflt / best, 4)}
        rows.append(row)
        if a.out:                                                       # (after every regime: a run cut short keeps what it has)
            json.dump({"tool": "tools/sweep_regimes.py", "budget_s_per_measurement": a.budget_s, "regimes": len(rows), "partial": True,
                       "wall_s": round(time.time() - t_all, 1), "rows": rows}, open(a.out, "w"), indent=1)
        print(f"{g:8s} b={b:5d} n={n:5d} p={period:2d} R={r}: default {dflt:10.0f}  best forced {best_k:24s} {forced[best_k]:10.0f}  ratio {dflt / best:.3f}", flush=True)
    below = [r for r in rows if r["default_over_best"] < 0.97]
    out = {"tool": "tools/sweep_regimes.py", "budget_s_per_measurement": a.budget_s, "regimes": len(rows), "wall_s": round(time.time() - t_all, 1),
           "worst_default_over_best": min((r["default_over_best"] for r in rows), default=None),
           "regimes_below_0.97": [{k: r[k] for k in ("geometry", "batch_cif_equivalent", "period", "ranges", "default_fps", "best_forced", "default_over_best")} for r in below],
           "rows": rows}
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))


if __name__ == "__main__":
    main()
