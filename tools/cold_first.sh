#!/bin/bash
# Cold start where the driver measures it: icsp_enc as the FIRST GPU process of a fresh lease (this script must be the first thing
# a gpurun call does), then again, then beside a process that holds the device (what bench.py's e2e leg is), each with the program's
# --stats split and icsp_create's phases (ICSP_TRACE_CREATE); last a rocprofv3 --hip-trace --stats of one run.
# -> gpurun_out/<tag>/cold_first.txt ; usage: tools/cold_first.sh [tag] [frames]
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}; N=${2:-300}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
LOG=$OUT/cold_first_$N.txt; : > $LOG
T=/dev/shm/coldf_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
import numpy as np
c = clipgen.synth_clip('foremanlike', 300)
with open('foremanlike_cif(352X288)_${N}f.yuv', 'wb') as f:
    for k in range(($N + 299) // 300): f.write(c[: min(300, $N - 300 * k)].tobytes())"
run() {   # label, env...
  local label=$1; shift
  local s=$(date +%s.%N)
  env "$@" ICSP_TRACE_CREATE=1 $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_${N}f.yuv" -n $N -q 16 --intraPeriod 0 --stats > run.out 2> run.err
  local e=$(date +%s.%N)
  echo "== $label: wall $(echo "$e - $s" | bc) s  ->  $(echo "$N / ($e - $s)" | bc) frames/s" >> $LOG
  grep icsp_enc run.out | cut -c1-600 | sed 's/"worker0.*//' >> $LOG
  grep icsp_create run.err >> $LOG
}
run "first GPU process of the lease"
run "second process"
run "third process"
# a process that holds the device while icsp_enc runs (bench.py's situation)
python3 -c "
import torch, time
x = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize(); print('holding', flush=True); time.sleep(8)" > hold.out 2>&1 &
HP=$!
sleep 4
run "beside a process that holds the device"
run "beside a process that holds the device, again"
wait $HP
sleep 1
run "device idle again"
export TMPDIR=/tmp
rocprofv3 --hip-trace --stats -d $OUT/cold_hip_$N --output-format csv -- $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_${N}f.yuv" -n $N -q 16 --intraPeriod 0 --stats > hip.log 2>&1
python3 - >> $LOG <<PY
import csv, glob
print("== rocprofv3 --hip-trace --stats (one run):")
for p in glob.glob("$OUT/cold_hip_$N/**/*hip_api_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:16]:
        print(f"{r['Name'][:44]:44s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
cat $LOG
cd /; rm -rf $T; find $OUT/cold_hip_$N -name "*.csv" -size +1M -delete
