#!/bin/bash
# EO_PIPE_DDEPTH: prefetch distance of the dY image of a pipeline stage (2 against the X image's 3): parity, then same box, alternating.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o|pipe_abl')
mk() { $HC $2 -c eonerf_bwd_pipe.hip -o build/pipe_vd$1.o && $HC -shared -o build/libeonerf_vd$1.so $REST build/pipe_vd$1.o || exit 1; }
mk 3 "-DEO_PIPE_DDEPTH=3"

mk 2 "-DEO_PIPE_DDEPTH=2"
cd ../..
B=$PWD/eonerf_code_amd/csrc/build
EONERF_LIB=$B/libeonerf_vd2.so timeout -k 10 400 python -m pytest tests/test_bwd_pipe.py -m gpu -q -x 2>&1 | tail -3
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
  run "dY 3 steps ahead (as the X image) " $B/libeonerf_vd3.so
  run "dY 2 steps ahead                  " $B/libeonerf_vd2.so

done
