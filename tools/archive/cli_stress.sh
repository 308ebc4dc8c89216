#!/bin/bash
# icsp_enc many times over (bounded by timeout): any hang or wrong output?  tools/cli_stress.sh [runs]  (GPU box, repo root)
N=${1:-100}
T=$(mktemp -d -p /dev/shm)
python3 - "$T" <<'PY'
import sys
sys.path.insert(0, ".")
from icspcodec_amd import clipgen
import numpy as np
c = clipgen.synth_clip("foremanlike", 300)
np.concatenate([c] * 10).tofile(sys.argv[1] + "/long_cif(352X288)_3000f.yuv")
PY
R=$GRAFT_REPO_ROOT
cd "$T"
ok=0; bad=0; ref=""
for i in $(seq 1 $N); do
  s=$((1 + i % 4)); p=$(( (i % 3) * 5 )); [ $p -eq 5 ] && p=10
  if timeout 20 $R/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 8 --intraPeriod 10 --streams $s --chunk $((250 + 125 * (i % 7))) > out.txt 2>&1; then
    h=$(cat long_compCIF_8_8_10.bin test_yuv.yuv | sha256sum | cut -c1-16)
    [ -z "$ref" ] && ref=$h
    if [ "$h" = "$ref" ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i: output differs ($h vs $ref)"; fi
  else bad=$((bad+1)); echo "run $i streams $s: rc=$? (124 = timeout)"; tail -1 out.txt | cut -c1-200; fi
done
echo "ok=$ok bad=$bad ref=$ref"
cd $R; rm -rf "$T"
