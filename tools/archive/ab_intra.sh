#!/bin/bash
# A/B of the 8-lane intra kernel's wavefront forms on the GPU box (through gpurun, from the repo root): the plain wavefront
# (ICSP_INTRA_GROUP=1) against rows chained in fours (4), per regime.  Usage: tools/ab_intra.sh [passes]
P=${1:-200}
for rep in 1; do
for g in 1 2 4; do
  echo "== ICSP_INTRA_GROUP=$g"
  ICSP_INTRA_GROUP=$g python tools/alt_ranges.py 0 16 300 2 $P
  ICSP_INTRA_GROUP=$g python tools/alt_ranges.py 0 16 300 1 $P
  ICSP_INTRA_GROUP=$g ICSP_INTRA_FORM=8 python tools/alt_ranges.py 0 16 100 1 $P
  ICSP_INTRA_GROUP=$g python tools/alt_ranges.py 0 16 3390 1 20
  ICSP_INTRA_GROUP=$g python tools/alt_ranges.py 10 8 300 2 $P
done
done
