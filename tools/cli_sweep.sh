#!/bin/bash
# icsp_enc end to end (file -> .bin + test_yuv.yuv) for several shard counts; run on the GPU box from the repo root
set -e
T=$(mktemp -d -p /dev/shm)
python3 - "$T" <<'PY'
import sys
sys.path.insert(0, ".")
from icspcodec_amd import clipgen
import numpy as np
c = clipgen.synth_clip("foremanlike", 300)
c.tofile(sys.argv[1] + "/" + clipgen.file_name("foremanlike", 300))
np.concatenate([c] * 10).tofile(sys.argv[1] + "/long_cif(352X288)_3000f.yuv")
PY
cd "$T"
for s in 1 2 4 8 16; do $GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --streams $s --stats | tail -1; done
for s in 4 8 16 32; do $GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 16 --intraPeriod 10 --streams $s --stats | tail -1; done
sha256sum foremanlike_compCIF_16_16_0.bin test_yuv.yuv | cut -c1-16
rm -rf "$T"
