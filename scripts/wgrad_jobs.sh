# Time of the weight-gradient GEMM launch per JOB (diagnostic: EONERF_WGRAD_MASK keeps only the named jobs, the gradients are wrong).
# Job order of the pipelined full step: 0 cam dY0 x enc, 1 cam dY5 x enc, 2 [dA1;dT1] x X8 (+ riders), 3 dA2 x A1, 4-6 dT2..4 x T1..3, 7 dT5 x T4,
# 8 sun dY0 x enc, 9 sun dY5 x enc, 10 sun d sigma x X8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wgrad_jobs; mkdir -p $O; cd $R
# the mask is compiled into a DIAGNOSTIC library only (-DEO_WGRAD_MASK); the shipped one ignores the variable
(cd eonerf_code_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DEO_WGRAD_MASK -c eonerf_api.hip -o build/api_mask.o \
  && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libeonerf_mask.so $(ls build/*.o | grep -v eonerf_api.o | grep -v api_mask.o | grep -v abl | grep -v stamp) build/api_mask.o) || exit 1
export EONERF_LIB=$R/eonerf_code_amd/csrc/build/libeonerf_mask.so
for M in 0x7ff 0x1 0x4 0x8 0x10 0x70 0x80 0x100 0x400 0x3 0x300 0xf8 0x704; do
  EONERF_WGRAD_MASK=$M python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload full 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('mask $M: wgrad %.4f ms  (step %.3f)' % (d['kernels']['wgrad_gemm']['avg_ms'], d['ms_per_step']))
"
done | tee $O/jobs.txt
