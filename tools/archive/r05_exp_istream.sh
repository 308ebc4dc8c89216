#!/bin/bash
# configs[2]'s regime (two IPPP ranges alternating): what the host pays per pass, and the I frames of every other range on a stream of
# their own (ICSP_I_STREAM_B=1, an experiment: a fourth busy stream of the context)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/exp_istream.txt
{
python tools/host_cost.py
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== ICSP_I_STREAM_B=$v rep $rep"
    ICSP_I_STREAM_B=$v python tools/alt_ranges.py 10 8 300 2 300
    ICSP_I_STREAM_B=$v python tools/alt_ranges.py 10 8 300 3 300
    ICSP_I_STREAM_B=$v python tools/alt_ranges.py 10 8 150 2 300
    ICSP_I_STREAM_B=$v python tools/alt_ranges.py 10 8 600 2 200
  done
done
echo "== ICSP_I_STREAM_B=1 GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 ICSP_I_STREAM_B=1 python tools/alt_ranges.py 10 8 300 2 300
echo "== ICSP_I_STREAM_B=0 GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 ICSP_I_STREAM_B=0 python tools/alt_ranges.py 10 8 300 2 300
} > $OUT 2>&1
cat $OUT | cut -c1-200
