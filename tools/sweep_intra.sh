# crossover of the intra luma kernel's two forms on all-intra CIF batches (through gpurun from the repo root):
# frames per pass x {32-lane, 8-lane}
cd $GRAFT_REPO_ROOT
{
for n in 400 450 512 600 800 1000 1500 3390; do
  p=$(( 60000 / n + 5 ))
  ICSP_INTRA_FORM=32 python tools/alt_ranges.py 0 16 $n 1 $p
  ICSP_INTRA_FORM=8 python tools/alt_ranges.py 0 16 $n 1 $p
done
for nr in "300 2" "200 2" "250 2" "300 3"; do
  ICSP_INTRA_FORM=32 python tools/alt_ranges.py 0 16 $nr 200
  ICSP_INTRA_FORM=8 python tools/alt_ranges.py 0 16 $nr 200
  python tools/alt_ranges.py 0 16 $nr 200
done
} 2>&1 | awk '{print $1,$3,$4,$5,$6,$7,$NF}' > gpurun_out/sweep_intra.txt
