#!/bin/bash
# Co-residency falsifier: can HALF a pipeline stage (a dX-only workgroup at 128 registers, 96 KB of LDS) share its CU with a
# streaming workgroup, and what does each of them then run at?  Diagnostic builds (gradients are WRONG in all but the first), same box,
# one call: bash scripts/coresidency.sh > gpurun_out/<dir>/coresidency.txt
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|eonerf_api.o|_v[0-9a-z]*\.o')
build() {  # name, pipe flags, api flags
  $HC $2 -c eonerf_bwd_pipe.hip -o build/pipe_v$1.o || exit 1
  $HC $3 -c eonerf_api.hip -o build/api_v$1.o || exit 1
  $HC -shared -o build/libeonerf_v$1.so $REST build/pipe_v$1.o build/api_v$1.o || exit 1
}
build dx "-DEO_PABL=128" ""
build dw "-DEO_PABL=50" ""
build cor "-DEO_COR=1 -DEO_PABL=128" "-DEO_COR=1"
cd ../..
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  for i in 1 2; do
    env "$@" EONERF_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload full 2> /tmp/cor.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); k = d['kernels']
print('$label run $i: step %.3f ms | pipe_cam %.4f pipe_sun %.4f wgrad %.4f heads %.4f fwd %.4f+%.4f' % (d['ms_per_step'], k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms'], k['wgrad_gemm']['avg_ms'], k['bwd_chain_camera']['avg_ms'], k['fwd_chain_camera']['avg_ms'], k['fwd_chain_sun']['avg_ms']))"
    grep '^\[cor\]' /tmp/cor.err | sed "s/^/    $label run $i /"
  done
}
B=$PWD/eonerf_code_amd/csrc/build
run "A shipped (dX+dW, 255 regs, 128 KB)        " ""
run "B dX-only half stage (146 regs, 128 KB)     " $B/libeonerf_vdx.so
run "C dW-only half stage, upper bound (176 regs)" $B/libeonerf_vdw.so
run "D dX-only @128 regs, 96 KB, depth 2, alone  " $B/libeonerf_vcor.so
run "E D + streaming partner BESIDE it (2 GB)    " $B/libeonerf_vcor.so EONERF_COR_PARTNER=1 EONERF_COR_GB=2
run "E3 D + streaming partner BESIDE it (3 GB)   " $B/libeonerf_vcor.so EONERF_COR_PARTNER=1 EONERF_COR_GB=3
run "F D, partner ALONE in front of it (2 GB)    " $B/libeonerf_vcor.so EONERF_COR_PARTNER=2 EONERF_COR_GB=2
