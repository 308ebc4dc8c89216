cd $GRAFT_REPO_ROOT
{
for n in 190 200 210 220 230 240 250 260 270; do
  ICSP_EXP_T8=21 python tools/alt_ranges.py 0 16 $n 2 300
  ICSP_EXP_T8=10 python tools/alt_ranges.py 0 16 $n 2 300
done
for n in 220 250; do
  ICSP_EXP_T8=21 python tools/alt_ranges.py 0 16 $n 3 300
  ICSP_EXP_T8=10 python tools/alt_ranges.py 0 16 $n 3 300
done
} 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$(NF-3),$(NF-2),$(NF-1),$NF}' > gpurun_out/exp_cap.txt
