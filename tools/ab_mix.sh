#!/bin/bash
# RECORD of an experiment that was not kept: the slot rotation with a per-frame offset on top of the step
# (-DICSP_ROT_MIX=1 as tools/lib_mix1.so: + frame slot / 256; =2 as lib_mix2.so: + three hashed bits of the frame slot); neither the macro
# nor the libraries are in the tree any more.
cd $GRAFT_REPO_ROOT
for lib in default lib_mix1.so lib_mix2.so; do
  L=""; [ "$lib" != "default" ] && L=$GRAFT_REPO_ROOT/tools/$lib
  echo "== lib=$lib"
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 300 2 100
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 1000 1 30
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 3390 1 20
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
ICSP_LIB=$GRAFT_REPO_ROOT/tools/lib_mix2.so timeout 60 python -m pytest tests/test_gpu_intra8.py -x -q 2>&1 | tail -1
ICSP_LIB=$GRAFT_REPO_ROOT/tools/lib_mix1.so timeout 60 python -m pytest tests/test_gpu_intra8.py -x -q 2>&1 | tail -1
