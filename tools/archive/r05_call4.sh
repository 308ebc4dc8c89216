#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest4.txt 2>&1
tail -5 $O/pytest4.txt
{
for rep in 1 2; do
  python tools/alt_ranges.py 0 16 300 2 300
  python tools/alt_ranges.py 0 16 3390 1 30
  python tools/alt_ranges.py 10 16 3390 1 30
  python tools/alt_ranges.py 10 8 300 2 300
done
} 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab4.txt
cat $O/ab4.txt
