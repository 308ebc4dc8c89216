#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest5.txt 2>&1
tail -5 $O/pytest5.txt
{
for rep in 1 2; do
  for many in 0 1; do
    echo "== many=$many rep $rep"
    ICSP_ALT_MANY=$many python tools/alt_ranges.py 0 16 150 4 400
    ICSP_ALT_MANY=$many python tools/alt_ranges.py 0 16 300 3 300
    ICSP_ALT_MANY=$many python tools/alt_ranges.py 0 16 300 2 300
    ICSP_ALT_MANY=$many python tools/alt_ranges.py 10 8 150 4 400
    ICSP_ALT_MANY=$many python tools/alt_ranges.py 10 8 300 2 300
  done
done
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab_many.txt
cat $O/ab_many.txt
