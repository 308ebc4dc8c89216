#!/bin/bash
# where the pairs-chained wavefront (ICSP_INTRA_GROUP=2) wins: CIF all-intra, batch size x {one range again and again, two alternating}
# x {32-lane, 8-lane plain, 8-lane pairs, default}  (through gpurun)
for R in 2 1; do
for n in 100 150 200 250 270 300 350 400 500 600 800 1000 3390; do
  P=200; [ $n -ge 1000 ] && P=40
  a=$(ICSP_INTRA_FORM=32 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
  b=$(ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=1 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
  c=$(ICSP_INTRA_FORM=8 ICSP_INTRA_GROUP=2 python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
  d=$(python tools/alt_ranges.py 0 16 $n $R $P | awk '{print $5}')
  echo "R=$R n=$n  32-lane $a  8-lane $b  pairs $c  default $d"
done
done
