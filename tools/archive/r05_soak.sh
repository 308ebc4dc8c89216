#!/bin/bash
# tools/repro_fault.py, 150 s per mode (about a thousand rounds = eight thousand transfers each), this round's library, no mallopt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/repro_soak.txt
: > $OUT
for mode in plain reg regsplit; do
  echo "== mode=$mode" >> $OUT
  timeout 400 python3 tools/repro_fault.py $mode 150 7 >> $OUT 2> gpurun_out/r05/repro_soak_err.txt
  echo "exit=$?" >> $OUT
  grep -m2 -i "fault\|error\|abort" gpurun_out/r05/repro_soak_err.txt >> $OUT
done
cat $OUT
python -c "import __graft_entry__ as g; g.smoke()"
