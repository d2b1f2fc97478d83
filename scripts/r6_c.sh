O=gpurun_out/r6c; mkdir -p $O
TWIN_DENSE=1 timeout -k 10 300 python scripts/twin_converge.py 2000 6000 > $O/converge_dense.txt 2>&1; tail -8 $O/converge_dense.txt
timeout -k 10 900 python -m pytest tests/test_n_samples.py tests/test_dp_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
