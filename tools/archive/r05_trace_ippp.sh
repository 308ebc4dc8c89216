#!/bin/bash
# configs[2]'s regime (two IPPP ranges of 30 GOPs alternating) as a timeline: under rocprofv3 --kernel-trace (which then paces the run) and
# from the library's own HIP events (ICSP_TIMELINE_DUMP: no profiler attached)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
rm -rf /tmp/tr_ippp /tmp/tl_ippp.txt
rocprofv3 --kernel-trace -d /tmp/tr_ippp --output-format csv -- python3 tools/alt_ranges.py 10 8 300 2 60 > gpurun_out/r05/trace_ippp.log 2>&1
python3 tools/timeline_ippp.py /tmp/tr_ippp 1100 > gpurun_out/r05/timeline_rocprof_ippp.txt 2>&1
ICSP_TIMELINE_DUMP=/tmp/tl_ippp.txt python3 tools/alt_ranges.py 10 8 300 2 60 >> gpurun_out/r05/trace_ippp.log 2>&1
python3 tools/timeline_events.py /tmp/tl_ippp.txt 1000 0.8 > gpurun_out/r05/timeline_events_ippp.txt 2>&1
tail -3 gpurun_out/r05/timeline_rocprof_ippp.txt gpurun_out/r05/timeline_events_ippp.txt
