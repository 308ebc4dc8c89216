#!/bin/bash
# icsp_enc on 3000 CIF frames, three runs per setting, with the per-worker wall-clock split: worker count x chunk size x mode
# (the sweep behind the defaults: two workers per device, shared transfer streams, uploads in turn).  Run on the GPU box.
set -e
T=$(mktemp -d -p /dev/shm)
python3 - "$T" <<'PY'
import sys
sys.path.insert(0, ".")
from icspcodec_amd import clipgen
import numpy as np
c = clipgen.synth_clip("foremanlike", 300)
np.concatenate([c] * 10).tofile(sys.argv[1] + "/long_cif(352X288)_3000f.yuv")
c.tofile(sys.argv[1] + "/short_cif(352X288)_300f.yuv")
PY
cd "$T"
E=$GRAFT_REPO_ROOT/icspcodec_amd/icsp_enc
run() { "$@" --stats | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().split('[icsp_enc]',1)[1]); w=d['worker0']
print('   e2e %.0f  encode_s %.4f  w0: up %.4f count %.4f turn %.4f pack %.4f down %.4f  workers %d chunks %d init %.3f' % (d['e2e_fps_excl_init'], d['encode_s'], w['upload_s'], w['pack_count_s'], w['turn_wait_s'], w['pack_s'], w['download_s'], d['workers'], d['chunks'], d['init_s']))"; }
for mode in "-q 16 --intraPeriod 0" "-q 16 --intraPeriod 10" "-q 8 --intraPeriod 10"; do
for extra in "" "--streams 1" "--streams 3" "--streams 4 --chunk 256" "--chunk 256" "--chunk 1024"; do echo "mode $mode $extra"; for r in 1 2 3; do run $E -i "long_cif(352X288)_3000f.yuv" -n 3000 $mode $extra; done; done
done
rm -rf "$T"
