cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
{
for rep in 1 2 3; do
  for v in r04 early new; do
    lib=""; [ $v = r04 ] && lib=$GRAFT_REPO_ROOT/tools/lib_r04.so; [ $v = early ] && lib=$GRAFT_REPO_ROOT/tools/lib_early.so
    echo "== $v rep $rep"
    ICSP_LIB=$lib python tools/alt_ranges.py 0 16 300 2 300
    ICSP_LIB=$lib python tools/alt_ranges.py 0 16 3390 1 30
    ICSP_LIB=$lib python tools/alt_ranges.py 0 16 600 2 100
  done
done
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab7.txt
cat $O/ab7.txt
python -m pytest tests/test_gpu_intra8.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
