#!/bin/bash
# the chained form beside the one-per-CU chroma launch: LDS reservation sweep (through gpurun)
P=${1:-200}
for cap in 0 24 32 36 40 44 60; do
  echo "== rows chained, ICSP_CHROMA_CAP=$cap"
  ICSP_CHROMA_CAP=$cap ICSP_INTRA_GROUP=4 python tools/alt_ranges.py 0 16 300 2 $P
done
ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 300 2 $P
for n in 200 256 350 400 600; do
  ICSP_CHROMA_CAP=40 ICSP_INTRA_GROUP=4 python tools/alt_ranges.py 0 16 $n 2 $P
  ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 $n 2 $P
done
ICSP_CHROMA_CAP=40 ICSP_INTRA_GROUP=4 python tools/alt_ranges.py 10 8 300 2 $P
ICSP_CHROMA_CAP=40 ICSP_INTRA_GROUP=4 python tools/alt_ranges.py 0 16 300 3 $P
ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 300 3 $P
