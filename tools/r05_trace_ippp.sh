#!/bin/bash
# kernel trace of configs[2]'s regime (two IPPP ranges of 30 GOPs alternating) -> a steady-state timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for m in 0 1; do
  rm -rf /tmp/tr_$m
  ICSP_GRAPH=$m rocprofv3 --kernel-trace -d /tmp/tr_$m --output-format csv -- python3 tools/alt_ranges.py 10 8 300 2 60 > gpurun_out/r05/trace_ippp_g$m.log 2>&1
  python3 tools/timeline_ippp.py /tmp/tr_$m 1100 > gpurun_out/r05/timeline_ippp_g$m.txt 2>&1
done
head -120 gpurun_out/r05/timeline_ippp_g0.txt
tail -3 gpurun_out/r05/timeline_ippp_g1.txt
