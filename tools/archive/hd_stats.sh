#!/bin/bash
# per-kernel durations of the 1088p P step under load (BASELINE configs[4] geometry): through gpurun from the repo root
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/hd; rm -rf $OUT; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/hd_work.py <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np, time
from icspcodec_amd import capi, clipgen
n, w, h = 3000, 1920, 1088
base = clipgen.synth_clip("tablelike", 12, width=w, height=h)
enc = capi.Encoder(w, h, 16, 16, 30, max_frames=n)
for f in range(0, n, 12):
    enc.upload(base[:min(12, n - f)], first=f)
enc.encode_resident(0, n); enc.sync()
t0 = time.perf_counter()
for _ in range(3): enc.encode_resident(0, n)
enc.sync()
print("fps", n * 3 / (time.perf_counter() - t0))
enc.close()
PY
rocprofv3 --kernel-trace --stats -d $OUT/s --output-format csv -- python3 /tmp/hd_work.py > $OUT/run.txt 2>&1
cd $R
f=$(find $OUT/s -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > $OUT/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {r['Percentage']}%")
PY
tail -2 $OUT/run.txt >> $OUT/summary.txt
find $OUT/s -name "*kernel_trace.csv" -size +20M -delete
