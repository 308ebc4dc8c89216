#!/bin/bash
# P-step chains as graphs (ICSP_GRAPH 0 / 1 / 2): parity first, then the IPPP regimes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/exp_graph.txt
{
for m in 1 2; do
  echo "== parity ICSP_GRAPH=$m"
  ICSP_GRAPH=$m timeout 900 python -m pytest tests/test_gpu_ranges.py tests/test_gpu_parity.py tests/test_gpu_longgop.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
done
python tools/host_cost.py
for rep in 1 2; do
  for m in 0 1 2; do
    echo "== ICSP_GRAPH=$m rep $rep"
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 300 2 300
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 300 3 300
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 150 2 300
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 300 1 300
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 600 2 200
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 8 1200 2 100
    ICSP_GRAPH=$m python tools/alt_ranges.py 10 16 3390 1 30
    ICSP_GRAPH=$m ICSP_ALT_MANY=2 python tools/alt_ranges.py 10 8 150 4 300
    ICSP_GRAPH=$m python tools/hd_groups.py 2>/dev/null | tail -1
  done
done
} > $OUT 2>&1
cut -c1-150 $OUT
