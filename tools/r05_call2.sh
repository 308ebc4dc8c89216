#!/bin/bash
# round 5, second GPU call: the parity suite (half-frame units included), their A/B across loads, the fault reproducer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest2.txt 2>&1
tail -5 $O/pytest2.txt
{
for rep in 1 2 3; do
for v in whole split; do
  env=""; [ $v = split ] && env="ICSP_INTRA_SPLIT=1"
  echo "== $v rep $rep"
  env $env python tools/alt_ranges.py 0 16 300 2 300
  env $env python tools/alt_ranges.py 0 16 300 1 300
  env $env python tools/alt_ranges.py 0 16 250 2 300
  env $env python tools/alt_ranges.py 0 16 600 2 100
  env $env python tools/alt_ranges.py 0 16 1000 1 60
  env $env python tools/alt_ranges.py 0 16 3390 1 30
done
done
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab_split.txt
cat $O/ab_split.txt
timeout 1500 bash tools/repro_fault.sh 45
