#!/bin/bash
# icsp_enc on 3000 CIF frames under rocprofv3 --kernel-trace --memory-copy-trace: who moves what when (run on the GPU box)
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
OUT=$GRAFT_REPO_ROOT/gpurun_out/clitrace2
rm -rf $OUT; mkdir -p $OUT
cd "$T"
export TMPDIR=/tmp
for g in 1; do
timeout 120 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/g$g --output-format csv -- $GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 16 --intraPeriod ${PERIOD:-0} --stats > $OUT/g$g.log 2>&1 || echo "rc $?"
tail -1 $OUT/g$g.log | cut -c1-400
done
cd $GRAFT_REPO_ROOT
rm -rf "$T"
find $OUT -name "*agent_info*" -delete
