#!/bin/bash
# the host program on a 3000-frame CIF clip, --intraPeriod 10: one I stream (ICSP_I_STREAM_B=0) against one per chain
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/clab && cd /dev/shm/clab
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from icspcodec_amd import clipgen
c = clipgen.synth_clip("stefanlike", 300)
with open("long_cif(352X288)_3000f.yuv", "wb") as f:
    for _ in range(10): f.write(c.tobytes())
PY
for rep in 1 2 3; do for v in 0 1; do
  echo -n "ICSP_I_STREAM_B=$v: "
  ICSP_I_STREAM_B=$v /root/repo/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 8 --intraPeriod 10 --stats 2>&1 | grep "^\[icsp_enc\]" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()[len('[icsp_enc]'):]); print({k: d[k] for k in d if 'fps' in k or k in ('encode_s','workers','chunks')})"
done; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r05/cli_ab.txt
rm -rf /dev/shm/clab
