mkdir -p gpurun_out/r2c
timeout -k 10 400 python -m pytest tests/test_hip_fullsize.py tests/test_bwd_pipe.py -q -x > gpurun_out/r2c/det.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/det.log; tail -15 gpurun_out/r2c/det.log | cut -c1-300
