#!/bin/bash
# wall time of icsp_enc on a long clip (through gpurun): mapped + pinned files against --staged, each twice; 1.5 s between runs; "outside" = process start + what the kernel does after _exit
R=$GRAFT_REPO_ROOT
N=${1:-3000}
T=/dev/shm/coldl_$$; mkdir -p $T; cd $T
python3 -c "
import sys; sys.path.insert(0, '$R')
from icspcodec_amd import clipgen
c = clipgen.synth_clip('foremanlike', 300)
with open('long_cif(352X288)_${N}f.yuv', 'wb') as f:
    for k in range(($N + 299) // 300): f.write(c[: min(300, $N - 300 * k)].tobytes())"
one() {
  local label=$1; shift
  sleep 1.5
  python3 - "$label" "$@" <<PY
import subprocess, time, json, sys, os
env = dict(os.environ)
args = []
for a in sys.argv[2:]:
    if "=" in a and not a.startswith("-"): k, v = a.split("=", 1); env[k] = v
    else: args.append(a)
t0 = time.perf_counter()
r = subprocess.run(["$R/icspcodec_amd/icsp_enc", "-i", "long_cif(352X288)_${N}f.yuv", "-n", "$N", "-q", "16", "--stats"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
w = time.perf_counter() - t0
st = [l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")]
s = json.loads(st[0][10:]) if st else {}
inside = s.get('init_s', 0) + s.get('encode_s', 0) + s.get('bitstream_and_files_s', 0)
print(f"{sys.argv[1]:34s} wall {w:.3f} s = {$N / w:7.0f} fps  inside main {inside:.3f} (init {s.get('init_s')} hip {s.get('hip_start_s')} map {s.get('map_files_s')} pin {s.get('pin_mappings_s')} mapwait {s.get('setup_worker0', {}).get('wait_for_mappings_s')} encode {s.get('encode_s')} files {s.get('bitstream_and_files_s')}) outside {w - inside:.3f}")
PY
}
for rep in 1 2; do
for p in 0 10; do
one "mapped p=$p" --intraPeriod $p
one "staged p=$p" --intraPeriod $p --staged
one "mapped p=$p (again)" --intraPeriod $p
one "staged p=$p (again)" --intraPeriod $p --staged
done
done
cd /; rm -rf $T
