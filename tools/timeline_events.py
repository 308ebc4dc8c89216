#!/usr/bin/env python3
"""A window of the launch timeline the library wrote under ICSP_TIMELINE_DUMP=<file> (HIP events around every launch, no profiler
attached): timeline_events.py <file> [window_us] [start_fraction]"""
import sys
NAMES = {0: "intra_luma", 1: "chroma_dc", 2: "residual", 3: "me", 4: "frame_serial", 5: "pack", 6: "decode"}     # include/icsp_hip.h: ICSP_K_*
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 4: rows.append((float(p[2]), float(p[3]), int(p[1]), int(p[0])))
rows.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.75
t0 = rows[int(len(rows) * frac)][0]
sel = [r for r in rows if t0 <= r[0] < t0 + win]
SID = {0: "stream", 1: "stream2", 2: "pstream1", 3: "pstream2", 4: "other"}
last = {}
print(f"{len(rows)} launches; window of {win:.0f} us from {t0:.0f} us")
for s, e, sid, k in sel:
    gap = s - last[sid] if sid in last else float("nan")
    last[sid] = e
    print(f"{s - t0:9.1f} {e - s:8.1f} {gap:8.1f}  {SID.get(sid, sid):9s} {NAMES.get(k, k)}")
lum = [r for r in rows if NAMES.get(r[3], "") == "intra_luma"]
if len(lum) > 20:
    st = sorted(lum[i + 1][0] - lum[i][0] for i in range(len(lum) // 2, len(lum) - 1))
    print(f"luma I launches: median start-to-start {st[len(st) // 2]:.1f} us")
