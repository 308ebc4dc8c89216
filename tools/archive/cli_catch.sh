#!/bin/bash
# run icsp_enc (3000 CIF frames, --intraPeriod 10) under rocprofv3 --kernel-trace --memory-copy-trace until a run's streaming
# part takes more than 17 ms; keep that run's traces under gpurun_out/catch/ (run on the GPU box)
set -e
T=$(mktemp -d -p /dev/shm)
python3 - "$T" <<'PY'
import sys
sys.path.insert(0, ".")
from icspcodec_amd import clipgen
import numpy as np
c = clipgen.synth_clip("foremanlike", 300)
np.concatenate([c] * 10).tofile(sys.argv[1] + "/long_cif(352X288)_3000f.yuv")
PY
OUT=$GRAFT_REPO_ROOT/gpurun_out/catch
rm -rf $OUT; mkdir -p $OUT
cd "$T"
export TMPDIR=/tmp
for i in $(seq 1 ${RUNS:-24}); do
  rm -rf $OUT/run
  timeout 120 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/run --output-format csv -- $GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 16 --intraPeriod 10 --stats > $OUT/run.log 2>&1 || echo "rc $?"
  ms=$(grep -o '"encode_s": [0-9.]*' $OUT/run.log | head -1 | awk '{print $2*1000}')
  echo "run $i: encode $ms ms"
  if python3 -c "import sys; sys.exit(0 if float('$ms') > 17.0 else 1)"; then echo "caught"; mv $OUT/run $OUT/slow; cp $OUT/run.log $OUT/slow.log; break; fi
done
cd $GRAFT_REPO_ROOT
rm -rf "$T"
find $OUT -name "*agent_info*" -delete
