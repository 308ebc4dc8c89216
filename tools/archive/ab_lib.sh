# A/B of two builds of the library on the resident regimes: tools/ab_lib.sh <other lib under tools/> (through gpurun from the repo root)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OTHER=$GRAFT_REPO_ROOT/tools/${1:-lib_prev.so}
{
for rep in 1 2 3; do
for lib in "$OTHER" ""; do
  echo "== lib=${lib:-default}"
  ICSP_LIB=$lib python tools/alt_ranges.py 0 16 300 2 300
  ICSP_LIB=$lib python tools/alt_ranges.py 0 16 300 1 300
  ICSP_LIB=$lib python tools/alt_ranges.py 0 16 3390 1 30
  ICSP_LIB=$lib python tools/alt_ranges.py 10 8 300 2 300
  ICSP_LIB=$lib python tools/alt_ranges.py 10 8 300 1 300
  ICSP_LIB=$lib python tools/alt_ranges.py 10 16 3390 1 30
done
done
} 2>&1 | awk '/^==/{print; next} {print $1,$3,$4,$5,$6,$7}' > gpurun_out/ab_lib.txt
