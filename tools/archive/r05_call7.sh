cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
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
echo "== lists (this round's library)"
ICSP_ALT_MANY=2 python tools/alt_ranges.py 0 16 150 4 400
ICSP_ALT_MANY=4 python tools/alt_ranges.py 0 16 150 4 400
python tools/alt_ranges.py 0 16 150 4 400
ICSP_ALT_MANY=3 python tools/alt_ranges.py 0 16 300 3 300
python tools/alt_ranges.py 0 16 300 3 300
ICSP_ALT_MANY=2 python tools/alt_ranges.py 10 8 150 4 400
ICSP_ALT_MANY=4 python tools/alt_ranges.py 10 8 150 4 400
python tools/alt_ranges.py 10 8 150 4 400
ICSP_ALT_MANY=2 python tools/alt_ranges.py 10 8 300 4 200
python tools/alt_ranges.py 10 8 300 4 200
} 2>&1 | awk '/^==/{print; next} {print $1,$2,$3,$4,$5,$6,$7,$8}' > $O/ab7.txt
cat $O/ab7.txt
python -m pytest tests/test_gpu_ranges.py -m gpu -x -q 2>&1 | tail -3
