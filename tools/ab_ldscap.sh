#!/bin/bash
# RECORD of an experiment (the ICSP_EXP_LDS_TOTAL hook is not in the tree any more): workgroups of the pairs kernel per CU capped through an LDS reservation (ICSP_EXP_LDS_TOTAL bytes per workgroup: 60000 -> two
# per CU, 50000 -> three, 40000 -> four), two alternating ranges of 300 frames and 3390 frames
cd $GRAFT_REPO_ROOT
for v in 0 60000 50000 40000; do
  echo "== ICSP_EXP_LDS_TOTAL=$v"
  ICSP_EXP_LDS_TOTAL=$v python tools/alt_ranges.py 0 16 300 2 100
  ICSP_EXP_LDS_TOTAL=$v python tools/alt_ranges.py 0 16 3390 1 20
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
