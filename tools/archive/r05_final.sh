#!/bin/bash
# round 5's records: the profile round (kernel traces, counter passes, calibration), then the bench line with its defaults and with the driver's arguments
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python - <<'PY'
import json
for f in ("bench_default","bench_steps20"):
    try:
        d=json.loads(open("gpurun_out/r05/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["repeats"]["spread_pct"], "ippp", d["ippp"]["value"], "c4", d["config4"]["value"], d["config4"]["all_intra_loaded"]["value"], "c5", d["config5"]["value"], {k:v for k,v in d.get("small_ranges",{}).items() if k!="is"}, d["roofline"]["frac"], (d["roofline"].get("binding") or {}).get("frac"), all(d["parity"].values()))
    except Exception as e:
        print(f, "ERR", e)
PY
