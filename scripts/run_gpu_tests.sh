set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q > gpurun_out/r2a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/tests.log
tail -40 gpurun_out/r2a/tests.log
python scripts/bf16_vs_fp32.py 400 > gpurun_out/r2a/bf16.log 2>&1; tail -5 gpurun_out/r2a/bf16.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 3000 gpurun_out/r2a/bench.json; tail -5 gpurun_out/r2a/bench.err
