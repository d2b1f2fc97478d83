#!/bin/bash
# Is the pipelined trunk backward bound by its fabric traffic?  Diagnostic builds with the stage's memory traffic removed piece by piece
# (EO_PABL bits of eonerf_bwd_pipe.hip: 8 = no LDS-DMA refill behind the prologue, 16 = no ring / slab stores, 128 = no dW phase); timing only.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o')
for N in "$@"; do
  $HC -DEO_PABL=$N -c eonerf_bwd_pipe.hip -o build/pipe_vt$N.o || exit 1
  $HC -shared -o build/libeonerf_vt$N.so $REST build/pipe_vt$N.o || exit 1
done
cd ../..
B=$PWD/eonerf_code_amd/csrc/build
for N in 0 "$@" 0; do
  LIB=""; [ "$N" != "0" ] && LIB=$B/libeonerf_vt$N.so
  EONERF_LIB=$LIB timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload full 2> /dev/null | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.readline()); k = d['kernels']
    print('EO_PABL=$N: step %.3f ms | pipe_cam %.4f pipe_sun %.4f wgrad %.4f heads %.4f fwd %.4f+%.4f' % (d['ms_per_step'], k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms'], k['wgrad_gemm']['avg_ms'], k['bwd_chain_camera']['avg_ms'], k['fwd_chain_camera']['avg_ms'], k['fwd_chain_sun']['avg_ms']))
except Exception as e:
    print('EO_PABL=$N: bench failed', e)"
done
