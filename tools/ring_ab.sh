# A/B of the 8-lane intra kernel's reconstruction ring (ICSP_INTRA_RING): through gpurun from the repo root
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_intra8.py tests/test_gpu_ranges.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/ring_tests.txt
{
for rep in 1 2 3; do
for r in 0 1; do
  ICSP_INTRA_RING=$r python tools/alt_ranges.py 0 16 300 2 300
  ICSP_INTRA_RING=$r python tools/alt_ranges.py 0 16 3390 1 30
  ICSP_INTRA_RING=$r python tools/alt_ranges.py 0 16 1000 1 60
done
done
} > gpurun_out/ring_ab.txt 2>&1
