#!/bin/bash
# icsp_enc end to end (file -> .bin + test_yuv.yuv) for several shard counts / chunk sizes / modes; run on the GPU box from
# the repo root:  tools/cli_sweep.sh > gpurun_out/cli_sweep.txt
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
E=$GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc
run() { "$@" --stats | tail -1 | cut -c1-1200; }
echo "== 300 f all-intra"
for s in 1 2; do for m in "" "--staged"; do echo "streams $s $m"; run $E -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --streams $s $m; done; done
sha256sum foremanlike_compCIF_16_16_0.bin test_yuv.yuv | cut -c1-16
echo "== 3000 f IPPP"
for s in 1 2 3 4; do for c in 250 500 1000; do echo "streams $s chunk $c"; run $E -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 8 --intraPeriod 10 --streams $s --chunk $c; done; done
sha256sum long_compCIF_8_8_10.bin test_yuv.yuv | cut -c1-16
echo "== 3000 f IPPP staged"
for s in 2 4; do echo "streams $s staged"; run $E -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 8 --intraPeriod 10 --streams $s --chunk 250 --staged; done
sha256sum long_compCIF_8_8_10.bin test_yuv.yuv | cut -c1-16
echo "== 3000 f IPPP, one GOP group per context"
for s in 2 4; do echo "streams $s P_GROUPS=1"; ICSP_P_GROUPS=1 run $E -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 8 --intraPeriod 10 --streams $s --chunk 500; done
echo "== 3000 f all-intra"
for s in 1 2 4; do echo "streams $s"; run $E -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 16 --intraPeriod 0 --streams $s; done
rm -rf "$T"
