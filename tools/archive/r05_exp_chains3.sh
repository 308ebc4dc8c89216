#!/bin/bash
# three chain streams when three (or more) all-intra ranges rotate (ICSP_CHAINS3=1): parity, then the rotation regimes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/exp_chains3.txt
{
echo "== parity ICSP_CHAINS3=1"
ICSP_CHAINS3=1 timeout 900 python -m pytest tests/test_gpu_ranges.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
  for v in 0 1; do
    echo "== ICSP_CHAINS3=$v rep $rep"
    for a in "150 4 400" "150 3 300" "100 3 300" "100 4 400" "200 3 300" "300 3 300" "300 4 300" "300 2 300" "600 3 200" "75 4 400"; do ICSP_CHAINS3=$v python tools/alt_ranges.py 0 16 $a | cut -c1-78; done
  done
done
} > $OUT 2>&1
cat $OUT
