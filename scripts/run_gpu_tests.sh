set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q -s > gpurun_out/r2a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/tests.log
grep -E "passed|failed|FAILED|worst exactness|bf16 vs fp32" gpurun_out/r2a/tests.log | cut -c1-900
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 4000 gpurun_out/r2a/bench.json; tail -8 gpurun_out/r2a/bench.err
