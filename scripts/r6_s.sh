O=gpurun_out/r6s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bwd_pipe.py tests/test_hip_backward.py tests/test_oracle_fullsize.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
run() {
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload full 2> $O/err.txt | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.readline()); k = d['kernels']
    print('$1: full %.3f ms (blocks %s) | pipe_cam %.4f pipe_sun %.4f tail/pair %.4f wgrad %.4f heads %.4f' % (d['ms_per_step'], ' '.join('%.3f' % b for b in d['blocks_ms_per_step']), k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms'], k['ig_tail_sun']['avg_ms'], k['wgrad_gemm']['avg_ms'], k['bwd_chain_camera']['avg_ms']))
except Exception as e:
    print('$1: failed', e)" || tail -3 $O/err.txt
}
for i in 1 2 3; do
  run EONERF_ENC_PAIR=0
  run EONERF_ENC_PAIR=1
done 2>&1 | tee $O/ab.txt
timeout -k 10 300 python -m pytest tests/test_n_samples.py -m gpu -q -k "backward_fp32" 2>&1 | tail -3
