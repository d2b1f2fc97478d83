# ready weight-gradient GEMM items on streaming workgroups of the pipelined CAMERA launch (EONERF_PIPE_STREAM = number of
# workgroups; the launch keeps (256 - that) / 7 pipelines).  One box, alternating runs, conditioning phase on; per-kernel event scopes of
# the pipelined camera launch (stages + streaming roles) and of the GEMM launch that takes what is left.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stream_ab; mkdir -p $O; cd $R
for rep in 1 2; do
for S in ${SWEEP:-0 4 18 32 46}; do
  EONERF_PIPE_STREAM=$S python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload ${WL:-full} 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
k = d['kernels']
print('stream $S: %.3f ms/step (blocks %s) | pipe_cam %.3f wgrad %.3f pipe_sun %.3f heads %.3f fwd %.3f+%.3f' % (d['ms_per_step'], ' '.join('%.3f' % b for b in d['blocks_ms_per_step']),
      k['bwd_pipe_camera']['avg_ms'], k['wgrad_gemm']['avg_ms'], k.get('bwd_pipe_sun', {}).get('avg_ms', 0), k['bwd_chain_camera']['avg_ms'], k['fwd_chain_camera']['avg_ms'], k.get('fwd_chain_sun', {}).get('avg_ms', 0)))
"
done
done | tee $O/ab_${WL:-full}.txt
