#!/bin/bash
# A/B/C of builds of the library on the resident regimes (through gpurun from the repo root): tools/ab_libs.sh lib1.so lib2.so ...
# ("default" = the shipped build); three interleaved repetitions
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in "$@"; do
  L=""; [ "$lib" != "default" ] && L=$GRAFT_REPO_ROOT/tools/$lib
  echo "== lib=$lib"
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 300 2 300
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 250 2 300
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 3390 1 30
  ICSP_LIB=$L python tools/alt_ranges.py 10 8 300 2 300
  ICSP_LIB=$L python tools/alt_ranges.py 10 16 3390 1 30
  ICSP_LIB=$L python tools/hd_groups.py 2>/dev/null | tail -1
done
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}'
