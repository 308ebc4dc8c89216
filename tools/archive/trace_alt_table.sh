#!/bin/bash
# kernel timeline of the alternating all-intra regime (through gpurun, from the repo root): start / duration / gap per queue of
# the last passes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_ai
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/alt_ranges.py ${1:-0} ${2:-16} 300 2 40 > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY' > $OUT/table.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
rows.sort()
rows = rows[-60:]
t0 = rows[0][0]
def short(n):
    n = n.split("(")[0]
    for p in ("(anonymous namespace)::", "void "): n = n.replace(p, "")
    return n[:30]
prev = {}
print(f"{'start_us':>9} {'end_us':>9} {'dur_us':>8} {'gap_us':>7} q  grid      kernel")
for s, e, n, q, g in rows:
    gap = (s - prev[q]) / 1e3 if q in prev else 0.0
    prev[q] = e
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:7.2f} {q:>2} {g:>9} {short(n)}")
PY
find $OUT -name "*.csv" -size +2M -delete
