#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
tail -15 $O/profile_round.log
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python - <<'PY'
import json
for f in ("bench_default","bench_steps20"):
    try:
        d=json.loads(open("gpurun_out/r05/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["repeats"], "ippp", d["ippp"]["value"], "c4", d["config4"]["value"], d["config4"]["all_intra_loaded"]["value"], "c5", d["config5"]["value"], d.get("small_ranges"), d["roofline"].get("binding"), d["parity"])
    except Exception as e:
        print(f, "ERR", e)
PY
{
for rep in 1 2 3; do
  for v in r04 new; do
    lib=""; [ $v = r04 ] && lib=$GRAFT_REPO_ROOT/tools/lib_r04.so
    echo "== $v rep $rep"
    ICSP_LIB=$lib python tools/alt_ranges.py 0 16 300 2 300
    ICSP_LIB=$lib python tools/alt_ranges.py 0 16 3390 1 30
  done
done
echo "== lists"
ICSP_ALT_MANY=2 python tools/alt_ranges.py 0 16 150 4 400
ICSP_ALT_MANY=4 python tools/alt_ranges.py 0 16 150 4 400
python tools/alt_ranges.py 0 16 150 4 400
ICSP_ALT_MANY=3 python tools/alt_ranges.py 0 16 300 3 300
python tools/alt_ranges.py 0 16 300 3 300
ICSP_ALT_MANY=2 python tools/alt_ranges.py 10 8 150 4 400
python tools/alt_ranges.py 10 8 150 4 400
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab6.txt
cat $O/ab6.txt
