#!/usr/bin/env python3
"""Timeline table from a rocprofv3 --kernel-trace directory: the kernels of the LAST pass of tools/trace_step.py."""
import csv, glob, os, sys
d = sys.argv[1]
npass_kernels = None
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
rows.sort()
# a pass starts at each k_intra_luma32 launch
starts = [i for i, r in enumerate(rows) if "k_intra_luma32" in r[2]]
lo = starts[-1]
# chroma kernels of the same pass may start slightly before/after; include everything from lo - 0 onward
last = rows[lo:]
t0 = last[0][0]
def short(n):
    n = n.split("(")[0]
    for p in ("(anonymous namespace)::", "void "):
        n = n.replace(p, "")
    return n[:34]
prev_end = {}
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} q  grid      kernel")
for s, e, n, q, g in last:
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:7.2f} {q:>2} {g:>9} {short(n)}")
print(f"pass total: {(max(r[1] for r in last) - t0) / 1e3:.2f} us")
