#!/bin/bash
# Is the slow first run of icsp_enc on a fresh box the page cache (the runtime's shared libraries read from disk)?  First thing
# of a gpurun call: read the libraries the program maps, then run it.  (tools/cold_first.sh is the same without the read.)
R=$GRAFT_REPO_ROOT
T=/dev/shm/coldc_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
clipgen.synth_clip('foremanlike', 300).tofile('foremanlike_cif(352X288)_300f.yuv')"
LIBS=$(ldd $R/icspcodec_amd/icsp_enc $R/icspcodec_amd/libicsp_hip.so | awk '/=> \//{print $3}' | sort -u)
echo "$LIBS" | tr '\n' ' '; echo
s=$(date +%s%N)
if [ "${1:-read}" = "read" ]; then for l in $LIBS; do cat $(readlink -f $l) > /dev/null; done; fi
e=$(date +%s%N); echo "reading the libraries: $(( (e - s) / 1000000 )) ms  ($(du -cLh $LIBS | tail -1))"
for i in 1 2 3; do
  s=$(date +%s%N)
  ICSP_TRACE_CREATE=1 $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --stats > o.txt 2> e.txt
  e=$(date +%s%N)
  echo "run $i: wall $(( (e - s) / 1000000 )) ms; $(grep -o '"init_s[^}]*}' o.txt | cut -c1-160)"; grep "stream\|attributes" e.txt
done
cd /; rm -rf $T
