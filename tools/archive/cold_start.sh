#!/bin/bash
# Cold start of icsp_enc on the 300-frame clip (through gpurun, from the repo root): wall time of the process, the program's own
# split, and the HIP API calls that cost the most (rocprofv3 --hip-trace, the program directly after --).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/cold
mkdir -p $OUT
T=/dev/shm/cold_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
clipgen.synth_clip('foremanlike', 300).tofile('foremanlike_cif(352X288)_300f.yuv')"
for i in 1 2 3 4 5 6; do
  s=$(date +%s.%N); $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --stats "$@" | grep icsp_enc > $OUT/run$i.txt; e=$(date +%s.%N)
  echo "wall $(echo "$e - $s" | bc) $(cut -c1-420 $OUT/run$i.txt | sed 's/"worker0.*//')"
done
sha256sum foremanlike_compCIF_16_16_0.bin test_yuv.yuv
export TMPDIR=/tmp
rocprofv3 --hip-trace --stats -d $OUT/hip --output-format csv -- $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --stats "$@" > $OUT/hip.log 2>&1
python3 - <<PY
import csv, glob
for p in glob.glob("$OUT/hip/**/*hip_api_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:14]:
        print(f"{r['Name'][:44]:44s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
cd /; rm -rf $T; find $OUT -name "*.csv" -size +2M -delete
