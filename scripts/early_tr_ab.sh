#!/bin/bash
# Transposed LDS reads of the pipeline stage issued ahead of their use (EO_PIPE_EARLY_TR, eonerf_bwd_pipe.hip), same box, alternating.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o')
mk() { $HC $2 -c eonerf_bwd_pipe.hip -o build/pipe_ve$1.o && $HC -shared -o build/libeonerf_ve$1.so $REST build/pipe_ve$1.o || exit 1; }
mk dw "-DEO_PIPE_EARLY_TR=1"
mk dx14 "-DEO_PIPE_EARLY_TR=2 -DEO_PIPE_XM_AT=14"
mk both14 "-DEO_PIPE_EARLY_TR=3 -DEO_PIPE_XM_AT=14"
mk both13 "-DEO_PIPE_EARLY_TR=3 -DEO_PIPE_XM_AT=13"
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
for i in 1 2 3; do
  run "base                 " ""
  run "dW reads early       " $B/libeonerf_vedw.so
  run "dX reads at MFMA 14  " $B/libeonerf_vedx14.so
  run "both (14)            " $B/libeonerf_veboth14.so
  run "both (13)            " $B/libeonerf_veboth13.so
done
EONERF_LIB=$B/libeonerf_veboth14.so timeout -k 10 300 python -m pytest tests/test_bwd_pipe.py -m gpu -q -x 2>&1 | tail -3
