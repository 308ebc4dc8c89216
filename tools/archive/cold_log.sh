#!/bin/bash
# where the HIP runtime spends its start-up: AMD_LOG_LEVEL=4 time stamps of one icsp_enc run (through gpurun)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/coldlog; mkdir -p $OUT
T=/dev/shm/cl_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
clipgen.synth_clip('foremanlike', 300).tofile('foremanlike_cif(352X288)_300f.yuv')"
for i in 1 2 3; do
AMD_LOG_LEVEL=4 $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --stats > $OUT/log$i.txt 2>&1
done
for i in 1 2 3; do
AMD_DIRECT_DISPATCH=0 AMD_LOG_LEVEL=4 $R/icspcodec_amd/icsp_enc -i "foremanlike_cif(352X288)_300f.yuv" -n 300 -q 16 --intraPeriod 0 --stats > $OUT/logdd$i.txt 2>&1
done
python3 - <<PY
import re, glob
for p in sorted(glob.glob("$OUT/log*.txt")):
    ts = []
    for ln in open(p, errors="replace"):
        m = re.search(r"us:\s*(\d+)", ln)
        if not m:
            m = re.search(r":(\d{6,}) us", ln)
        if m:
            ts.append((int(m.group(1)), ln.strip()[:170]))
    print("==", p.split("/")[-1], "lines with stamps:", len(ts))
    if not ts: 
        print(open(p, errors="replace").read()[:600]); continue
    gaps = sorted(((ts[k + 1][0] - ts[k][0], k) for k in range(len(ts) - 1)), reverse=True)[:6]
    print("   span ms", (ts[-1][0] - ts[0][0]) / 1e3)
    for g, k in sorted(gaps, key=lambda x: x[1]):
        print(f"   gap {g/1e3:8.1f} ms after: {ts[k][1][:150]}")
        print(f"                      next: {ts[k+1][1][:150]}")
    st = [l for l in open(p, errors="replace") if l.startswith("[icsp_enc]")]
    print("   ", st[0][:330] if st else "no stats")
PY
cd /; rm -rf $T; for f in $OUT/*.txt; do head -c 300000 $f > $f.cut; mv $f.cut $f; done
