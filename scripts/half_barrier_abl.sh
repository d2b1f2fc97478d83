#!/bin/bash
# Falsifier for a 64-sample (two sub-step) stage of the pipelined backward: EO_PABL=512 runs the workgroup barrier and the flag work of a
# stage on every second step only (results are garbage; timing only).  Same box, alternating with the shipped build.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o|pipe_abl')
mk() { $HC $2 -c eonerf_bwd_pipe.hip -o build/pipe_vh$1.o && $HC -shared -o build/libeonerf_vh$1.so $REST build/pipe_vh$1.o || exit 1; }
mk half "-DEO_PABL=512"
mk spread "-DEO_PIPE_SPREAD=1"
cd ../..
B=$PWD/eonerf_code_amd/csrc/build
run() {
  EONERF_LIB=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload full 2> /dev/null | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.readline()); k = d['kernels']
    print('$1: full %.3f ms (blocks %s) | pipe_cam %.4f pipe_sun %.4f' % (d['ms_per_step'], ' '.join('%.3f' % b for b in d['blocks_ms_per_step']), k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms']))
except Exception as e:
    print('$1: failed', e)"
}
for i in 1 2; do
  run "base                      " ""
  run "barrier every 2nd step    " $B/libeonerf_vhhalf.so
  run "DMA pieces between MFMAs  " $B/libeonerf_vhspread.so
done
