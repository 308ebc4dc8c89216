#!/bin/bash
# rotation of the plain wavefront's slots on frames that have no pairs form (720p, 1088p) and on the IPPP legs
cd $GRAFT_REPO_ROOT
for lib in default lib_rotplain.so default lib_rotplain.so; do
  L=""; [ "$lib" != "default" ] && L=$GRAFT_REPO_ROOT/tools/$lib
  echo "== lib=$lib"
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 60 2 40 1280 720
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 300 1 10 1280 720
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 30 2 30 1920 1088
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 165 1 10 1920 1088
  ICSP_LIB=$L python tools/hd_groups.py 2>/dev/null | tail -1
  ICSP_LIB=$L python tools/alt_ranges.py 10 16 3390 1 30
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
