#!/bin/bash
# full default-against-forced-knobs sweep of one or more geometries: tools/r05_sweep.sh CIF,352x576  -> gpurun_out/r05/sweep_<geoms>.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
G=${1:-CIF}
timeout ${2:-3000} python tools/sweep_regimes.py --geoms $G --budget-s 0.08 --out gpurun_out/r05/sweep_${G//,/_}.json > gpurun_out/r05/sweep_${G//,/_}.log 2>&1
tail -25 gpurun_out/r05/sweep_${G//,/_}.log
