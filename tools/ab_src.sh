#!/bin/bash
# RECORD of an experiment that was not kept ("source rows through the ring"): the build with the source rows
# going through the reconstruction ring as the default library against the build before it as tools/lib_prev.so (every task loads
# its own 8 bytes per row).  Neither library is in the tree any more; the script documents what was compared and how.
cd $GRAFT_REPO_ROOT
for lib in default lib_prev.so default lib_prev.so; do
  L=""; [ "$lib" != "default" ] && L=$GRAFT_REPO_ROOT/tools/$lib
  echo "== lib=$lib"
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 300 2 100
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 300 1 100
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 1000 1 30
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 3390 1 30
  ICSP_LIB=$L ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 3390 1 30
  ICSP_LIB=$L python tools/alt_ranges.py 10 8 300 2 100
  ICSP_LIB=$L python tools/alt_ranges.py 0 16 75 2 30 704 576
done 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
