#!/bin/bash
# EO_PIPE_DMA03: all LDS-DMA pieces of a pipeline step issued by waves 0-3.  Same box, alternating.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o')
$HC -DEO_PIPE_DMA03=1 -c eonerf_bwd_pipe.hip -o build/pipe_vd03.o && $HC -shared -o build/libeonerf_vd03.so $REST build/pipe_vd03.o || exit 1
$HC -DEO_PIPE_DMA03=1 -DEO_PIPE_STAMPS=1 -c eonerf_bwd_pipe.hip -o build/pipe_vd03s.o && $HC -shared -o build/libeonerf_vd03s.so $REST build/pipe_vd03s.o || exit 1
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
  run "shipped (every wave its 4 pieces)" ""
  run "waves 0-3 issue all 8            " $B/libeonerf_vd03.so
done
EONERF_LIB=$B/libeonerf_vd03.so timeout -k 10 300 python -m pytest tests/test_bwd_pipe.py -m gpu -q -x 2>&1 | tail -3
EONERF_LIB=$B/libeonerf_vd03s.so timeout -k 10 120 python scripts/pipe_stamps.py 2>&1 | grep -E "stage|L7 \| w[04]|L4 \| w[04]|L1 \| w[04]"
