#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of the bench command: how the launches of the dominant kernel overlap in the timed region.

    overlap_summary.py <dir with *kernel_trace.csv> <bench line (json file)> <out json>

bench.py's primary leg issues, on its first encoder, SETTLE_PASSES + warmup untimed steps and then repeats x steps timed ones, one
launch of the 8-lane luma kernel (300 workgroups) per step; the launches of the timed regions are picked by their place
in that sequence.  Per launch: start, end, queue, launches of the kernel in flight at its start; summary: medians of duration
and start-to-start distance (= the span of a step), and the chip-level rate that follows: algorithmic bytes of a launch / span."""
import csv, glob, json, os, re, sys
src, bench_json, out = sys.argv[1], sys.argv[2], sys.argv[3]
line = None
for l in open(bench_json):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        line = json.loads(l)
steps, warmup = line["steps"], line["warmup"]
repeats = line.get("n_repeats") or (line.get("repeats") or {}).get("repeats", 1)      # compact line (round 6) / the one long line of rounds 1-5
SETTLE = 100
P, NMB = 352 * 288, 396
BYTES = 300 * (4 * P + 8 * NMB)
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        # the 300-workgroup launches of the 8-lane luma kernel, whatever its variant (three waves, or four with the rows in pairs)
        if "k_intra_luma8" in r["Kernel_Name"] and int(r.get("Grid_Size") or r["Grid_Size_X"]) == 300 * int(r.get("Workgroup_Size") or r["Workgroup_Size_X"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
            kname = re.search(r"k_intra_luma8<[^>]*>", r["Kernel_Name"]).group(0)
rows.sort()
first = SETTLE + warmup
sel = rows[first: first + repeats * steps]
t0 = sel[0][0]
launches = []
for i, (s, e, q) in enumerate(sel):
    infl = sum(1 for (s2, e2, _) in rows if s2 <= s < e2)
    launches.append({"step": i, "start_us": round((s - t0) / 1e3, 2), "end_us": round((e - t0) / 1e3, 2), "dur_us": round((e - s) / 1e3, 2),
                     "queue": q, "in_flight_at_start": infl})
def med(x):
    x = sorted(x); return x[len(x) // 2]
durs = [l["dur_us"] for l in launches]
# start-to-start inside a region (the first launch of a region follows a host synchronisation)
gaps = [launches[i + 1]["start_us"] - launches[i]["start_us"] for i in range(len(launches) - 1) if (i + 1) % steps != 0]
span = med(gaps)
res = {"source": "rocprofv3 --kernel-trace of `bench.py --steps %d --warmup %d --repeats %d --no-cpu --legs ippp` (tools/profile_round.sh); "
                 "launches %d..%d of %s on 300 workgroups = the timed regions" % (steps, warmup, repeats, first, first + len(sel) - 1, kname),
       "launches": len(sel), "launch_duration_ms_median": round(med(durs) / 1e3, 4), "start_to_start_ms_median": round(span / 1e3, 4),
       "launches_in_flight_median": med([l["in_flight_at_start"] for l in launches]),
       "algorithmic_bytes_per_launch": BYTES,
       "per_launch_GBps_from_trace": round(BYTES / (med(durs) * 1e-6) / 1e9, 1),
       "chip_level_GBps_from_trace": round(BYTES / (span * 1e-6) / 1e9, 1),
       "bench_ms_per_step_same_run": line["ms_per_step"],
       "note": "duration > start-to-start distance: consecutive launches (independent batches on two streams) overlap; the span of a step "
               "is the start-to-start distance, and step bytes / span is the chip-level figure of roofline.chip_level (under the profiler)",
       "first_60_launches": launches[:60]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "first_60_launches"}, indent=1))
