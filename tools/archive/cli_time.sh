#!/bin/bash
# End-to-end wall time of the host programs on 300-frame CIF clips (process start, file I/O and HIP initialisation included),
# next to the reference CLI where oracle/_ref is built.  Run on the GPU box: gpurun -- 'bash tools/cli_time.sh'
set -e
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/clit && cd /tmp/clit
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from icspcodec_amd import clipgen
for name,n in (("foremanlike",300),("stefanlike",300)):
    clipgen.synth_clip(name,n).tofile(clipgen.file_name(name,n))
    print(clipgen.file_name(name,n))
PY
ls
F1=$(ls foremanlike*); F2=$(ls stefanlike*)
for i in 1 2; do t0=$(date +%s.%N); /root/repo/icspcodec_amd/icsp_enc -i "$F1" -n 300 -q 16 --intraPeriod 0 > /dev/null; t1=$(date +%s.%N); echo "icsp_enc all-intra 300f: $(python3 -c "print(round($t1 - $t0, 3))") s wall"; done
for i in 1 2; do t0=$(date +%s.%N); /root/repo/icspcodec_amd/icsp_enc -i "$F2" -n 300 -q 8 --intraPeriod 10 > /dev/null; t1=$(date +%s.%N); echo "icsp_enc ippp10 300f: $(python3 -c "print(round($t1 - $t0, 3))") s wall"; done
t0=$(date +%s.%N); /root/repo/icspcodec_amd/icsp_enc -i "$F2" -n 300 -q 8 --intraPeriod 10 --hostpack > /dev/null; t1=$(date +%s.%N); echo "icsp_enc ippp10 300f --hostpack: $(python3 -c "print(round($t1 - $t0, 3))") s wall"
if [ -x /root/repo/oracle/_ref/icsp_ref ]; then
t0=$(date +%s.%N); /root/repo/oracle/_ref/icsp_ref -i "$F1" -n 300 -q 16 --intraPeriod 0 > /dev/null; t1=$(date +%s.%N); echo "reference all-intra 300f: $(python3 -c "print(round($t1 - $t0, 3))") s wall"
t0=$(date +%s.%N); /root/repo/oracle/_ref/icsp_ref -i "$F2" -n 300 -q 8 --intraPeriod 10 > /dev/null; t1=$(date +%s.%N); echo "reference ippp10 300f: $(python3 -c "print(round($t1 - $t0, 3))") s wall"
fi
