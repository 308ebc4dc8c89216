R=$GRAFT_REPO_ROOT
T=/dev/shm/cs_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
c = clipgen.synth_clip('foremanlike', 300)
with open('long_cif(352X288)_3000f.yuv', 'wb') as f:
    for k in range(10): f.write(c.tobytes())"
for i in 1 2 3; do
sleep 1.5
ICSP_TRACE_CREATE=1 $R/icspcodec_amd/icsp_enc -i "long_cif(352X288)_3000f.yuv" -n 3000 -q 16 --intraPeriod 0 --stats 2>&1 | grep "icsp_copy_streams\|icsp_enc\]" | cut -c1-700 | sed 's/"worker0".*//'
done
cd /; rm -rf $T
