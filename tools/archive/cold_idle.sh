#!/bin/bash
# How icsp_enc's start-up depends on what the device did just before (through gpurun): delay since the previous GPU process
# exited; a second process holding the device idle; a second process keeping it busy.  Prints hip_start_s / create_s / wall.
set -u
R=$GRAFT_REPO_ROOT
T=/dev/shm/coldi_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
clipgen.synth_clip('foremanlike', 300).tofile('foremanlike_cif(352X288)_300f.yuv')"
one() {
  python3 - "$1" <<PY
import subprocess, time, json, sys
t0 = time.perf_counter()
r = subprocess.run(["$R/icspcodec_amd/icsp_enc", "-i", "foremanlike_cif(352X288)_300f.yuv", "-n", "300", "-q", "16", "--intraPeriod", "0", "--stats"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
w = time.perf_counter() - t0
st = [l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")]
s = json.loads(st[0][10:]) if st else {}
print(f"{sys.argv[1]:46s} wall {w:.3f} s  hip_start {s.get('hip_start_s')}  create {s.get('setup_worker0', {}).get('create_s')}  init {s.get('init_s')}")
PY
}
one "first process of the lease"
for d in 0.05 0.2 0.5 1 2 5; do sleep $d; one "after ${d}s since the previous exit"; done
echo "-- beside an idle holder (torch context, no work)"
python3 -c "
import torch, time
x = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize(); time.sleep(14)" &
HP=$!
sleep 3
for d in 0.05 0.5 1 2 3; do sleep $d; one "holder idle, ${d}s after previous exit"; done
wait $HP
echo "-- beside a holder that runs a tiny kernel every 5 ms"
python3 -c "
import torch, time
x = torch.zeros(1 << 10, device='cuda')
t = time.time()
while time.time() - t < 14:
    x += 1; torch.cuda.synchronize(); time.sleep(0.005)" &
HP=$!
sleep 3
for d in 0.05 0.5 1 2 3; do sleep $d; one "holder busy, ${d}s after previous exit"; done
wait $HP
cd /; rm -rf $T
