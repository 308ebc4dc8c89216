cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05
( time ICSP_FUZZ_SEEDS=1200 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 ) > $O/fuzz_default.txt 2>&1
tail -3 $O/fuzz_default.txt
( time ICSP_FUZZ_SEEDS=600 ICSP_INTRA_FORM=8 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 ) > $O/fuzz_form8.txt 2>&1
tail -3 $O/fuzz_form8.txt
( time ICSP_FUZZ_SEEDS=600 ICSP_INTRA_GROUP=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 ) > $O/fuzz_group1.txt 2>&1
tail -3 $O/fuzz_group1.txt
( time ICSP_FUZZ_SEEDS=300 ICSP_QUANT_POW2=0 ICSP_INTRA_FORM=8 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 ) > $O/fuzz_nopow2.txt 2>&1
tail -3 $O/fuzz_nopow2.txt
