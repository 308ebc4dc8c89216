#!/bin/bash
# round 5, first GPU call: the parity suite on the new transfers / k_me / quantiser, instruction counts, A/B of the builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest1.txt 2>&1
tail -5 $O/pytest1.txt
bash tools/insts_quick.sh new > /dev/null 2>&1
bash tools/insts_quick.sh r04 $GRAFT_REPO_ROOT/tools/lib_r04.so > /dev/null 2>&1
cat $O/insts_new.txt $O/insts_r04.txt
{
for rep in 1 2 3; do
for v in r04 metab new new_nopow2; do
  lib=""; env=""
  case $v in r04) lib=$GRAFT_REPO_ROOT/tools/lib_r04.so;; metab) lib=$GRAFT_REPO_ROOT/tools/lib_metab.so;; new_nopow2) env="ICSP_QUANT_POW2=0";; esac
  echo "== $v rep $rep"
  env ICSP_LIB=$lib $env python tools/alt_ranges.py 0 16 300 2 300
  env ICSP_LIB=$lib $env python tools/alt_ranges.py 0 16 3390 1 30
  env ICSP_LIB=$lib $env python tools/alt_ranges.py 10 16 3390 1 30
  env ICSP_LIB=$lib $env python tools/alt_ranges.py 10 8 300 2 300
  env ICSP_LIB=$lib $env python tools/alt_ranges.py 30 16 600 1 5 1920 1088
done
done
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab1.txt
cat $O/ab1.txt
