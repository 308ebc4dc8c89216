#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of tools/alt_ranges.py (IPPP, ranges alternating): the launches of a steady-state window as a
timeline -- per queue: kernel, start, duration, gap to the previous launch of the queue.

    timeline_ippp.py <dir with *kernel_trace.csv> [window_us] > text"""
import csv, glob, os, re, sys
src = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"k_\w+(<[^>]*>)?", n)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), m.group(0) if m else n[:30], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) // max(1, int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1))))
rows.sort()
# the middle of the run
t_mid = rows[len(rows) * 3 // 4][0]
sel = [r for r in rows if t_mid <= r[0] < t_mid + win * 1e3]
last_end = {}
print(f"{len(rows)} launches in the trace; window of {win:.0f} us from t = {t_mid}")
print(f"{'t_us':>9} {'dur_us':>8} {'gap_us':>8}  queue  kernel (workgroups)")
for s, e, q, k, wg in sel:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float('nan')
    last_end[q] = e
    print(f"{(s - t_mid) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {q:>5}  {k} ({wg})")
# busy time per queue in the window
from collections import defaultdict
busy = defaultdict(float)
for s, e, q, k, wg in sel: busy[q] += (e - s) / 1e3
print("busy us per queue in the window:", {q: round(v, 1) for q, v in busy.items()})
lum = [(s, e) for s, e, q, k, wg in rows if "intra_luma" in k]
if len(lum) > 20:
    d = sorted((e - s) / 1e3 for s, e in lum[len(lum) // 2:])
    st = sorted((lum[i + 1][0] - lum[i][0]) / 1e3 for i in range(len(lum) // 2, len(lum) - 1))
    print(f"luma I kernel: median duration {d[len(d) // 2]:.1f} us, median start-to-start {st[len(st) // 2]:.1f} us")
