set -x
mkdir -p gpurun_out/r2b
timeout -k 10 400 python -m pytest tests/test_bwd_pipe.py tests/test_hip_fullsize.py -x -q -s > gpurun_out/r2b/pipe_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pipe_tests.log
tail -25 gpurun_out/r2b/pipe_tests.log | cut -c1-400
timeout -k 10 200 python scripts/pipe_stamps.py > gpurun_out/r2b/stamps.log 2>&1; grep -E "w0|w4|pipelines" gpurun_out/r2b/stamps.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/bench_pipe.json 2> gpurun_out/r2b/bench_pipe.err; tail -3 gpurun_out/r2b/bench_pipe.err
