# GPU box: the whole -m gpu suite (one process), then the default bench run.  OUT=<dir under gpurun_out>
OUT=${OUT:-r3a}
mkdir -p gpurun_out/$OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -s --durations=15 > gpurun_out/$OUT/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$OUT/tests.log
grep -E "passed|failed|FAILED|Error|worst|fullsize|rc=" gpurun_out/$OUT/tests.log | cut -c1-600 | tail -40
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench.json 2> gpurun_out/$OUT/bench.err; tail -c 1500 gpurun_out/$OUT/bench.json; tail -8 gpurun_out/$OUT/bench.err
