#!/bin/bash
# RECORD of an experiment that was not kept (the kernel variant and its switch are not in the tree any more):
# the 32-lane intra kernel with the block rows in pairs (ICSP_INTRA32_PAIRS=1) against the default: parity first, then the regimes
# in which the 32-lane form is chosen (every frame on a CU of its own; the I step of an IPPP batch)
cd $GRAFT_REPO_ROOT
ICSP_INTRA32_PAIRS=1 timeout 75 python -m pytest tests/test_gpu_parity.py tests/test_geometry_golden.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do
  echo "== ICSP_INTRA32_PAIRS=$v"
  ICSP_INTRA32_PAIRS=$v python tools/alt_ranges.py 0 16 100 1 100
  ICSP_INTRA32_PAIRS=$v python tools/alt_ranges.py 10 8 300 2 100
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
