#!/bin/bash
# the step-wise rotation of slots over waves: pairs / fours / plain (lib_rotplain.so: the plain wavefront rotates too) across loads
cd $GRAFT_REPO_ROOT
for n in 300 600 1000 1500 2000 3390; do
  for R in 1 2; do
    P=100; [ $n -ge 1000 ] && P=30
    a=$(ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
    b=$(ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=2 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
    c=$(ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=4 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
    d=$(ICSP_LIB=$GRAFT_REPO_ROOT/tools/lib_rotplain.so ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
    e=$(ICSP_LIB=$GRAFT_REPO_ROOT/tools/lib_rotplain.so ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=1 ICSP_INTRA_NW=4 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
    echo "n=$n R=$R plain $a  pairs+rot $b  fours+rot $c  plain+rot $d  plain+rot,4 waves $e"
  done
done
