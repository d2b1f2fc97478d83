O=gpurun_out/r6g; mkdir -p $O
ab() { # label env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> $O/err.txt | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); k = d['kernels']; r = d.get('rgb', {})
print('$label: full %.3f ms (blocks %s) | pipe_cam %.4f pipe_sun %.4f wgrad %.4f heads %.4f fwd %.4f+%.4f | rgb %.3f ms' % (d['ms_per_step'], ' '.join('%.3f' % b for b in d['blocks_ms_per_step']), k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms'], k['wgrad_gemm']['avg_ms'], k['bwd_chain_camera']['avg_ms'], k['fwd_chain_camera']['avg_ms'], k['fwd_chain_sun']['avg_ms'], r.get('ms_per_step', 0)))" || tail -3 $O/err.txt
}
for i in 1 2; do
  ab "base      " EONERF_PIPE_XCD=0 EONERF_STAGGER=0
  ab "xcd       " EONERF_PIPE_XCD=1 EONERF_STAGGER=0
  ab "stagger1  " EONERF_PIPE_XCD=0 EONERF_STAGGER=1
  ab "stagger2  " EONERF_PIPE_XCD=0 EONERF_STAGGER=2
  ab "xcd+stag1 " EONERF_PIPE_XCD=1 EONERF_STAGGER=1
done 2>&1 | tee $O/ab.txt
EONERF_PIPE_XCD=1 EONERF_STAGGER=1 timeout -k 10 600 python -m pytest tests/test_bwd_pipe.py tests/test_hip_backward.py tests/test_hip_forward.py -m gpu -q -x > $O/tests_xcd_stag.log 2>&1; echo "rc=$?" >> $O/tests_xcd_stag.log; tail -4 $O/tests_xcd_stag.log
timeout -k 10 300 python scripts/dbg_ns2.py 96 > $O/dbg96.txt 2>&1; tail -36 $O/dbg96.txt
