#!/bin/bash
# after the I-stream change: the IPPP regimes with two and three ranges again (every geometry), then the 1088p regimes the first run did not reach
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout ${1:-2400} python tools/sweep_regimes.py --geoms CIF,352x576,4CIF,720p,1088p --periods 10 --ranges 2,3 --budget-s 0.08 --out gpurun_out/r05/sweep_ippp_r23.json > gpurun_out/r05/sweep_ippp_r23.log 2>&1
tail -12 gpurun_out/r05/sweep_ippp_r23.log
timeout ${2:-1200} python tools/sweep_regimes.py --geoms 1088p --batches 300,350,400,600,1000,3390 --periods 0,10 --ranges 1 --budget-s 0.08 --out gpurun_out/r05/sweep_1088p_rest_r1.json > gpurun_out/r05/sweep_1088p_rest_r1.log 2>&1
timeout ${3:-900} python tools/sweep_regimes.py --geoms 1088p --batches 300,350,400,600,1000,3390 --periods 0 --ranges 2,3 --budget-s 0.08 --out gpurun_out/r05/sweep_1088p_rest_ai.json > gpurun_out/r05/sweep_1088p_rest_ai.log 2>&1
tail -5 gpurun_out/r05/sweep_1088p_rest_r1.log gpurun_out/r05/sweep_1088p_rest_ai.log
