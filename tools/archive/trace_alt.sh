#!/bin/bash
# kernel trace of the alternating-ranges regime (through gpurun, from the repo root): trace_alt.sh <period> <qp> <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/alt_ranges.py $1 $2 300 2 60 > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
rows = []
for p in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# concurrency over the last 40 % of the run
t0 = int(rows[int(len(rows) * 0.6)]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > t0:
        ev.append((max(s, t0), 1)); ev.append((e, -1))
ev.sort()
busy = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
tot = sum(busy.values())
print("kernels in flight over the last 40 % of the run:", {k: round(v / tot, 3) for k, v in sorted(busy.items())})
PY
find $OUT -name "*.csv" -size +4M -delete
