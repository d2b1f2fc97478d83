#!/bin/bash
# XCD-local pipelines of the layer-pipelined backward (EONERF_PIPE_XCD=1), variants of the intra-XCD hand-off, same box, alternating.
#   plain0: roles by XCD only, hand-offs written through (sc1) as on a cross-XCD edge      nt: intra-XCD hand-offs loaded with the streaming policy
#   ringN: N ring slots in use (default 16)
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o')
mk() { $HC $2 -c eonerf_bwd_pipe.hip -o build/pipe_vx$1.o && $HC -shared -o build/libeonerf_vx$1.so $REST build/pipe_vx$1.o || exit 1; }
mk plain0 "-DEO_XCD_PLAIN=0"
mk nt "-DEO_XCD_NT=1"
mk ring8 "-DEO_RING_USE=8"
mk ring4nt "-DEO_RING_USE=4 -DEO_XCD_NT=1"
mk ring8nt "-DEO_RING_USE=8 -DEO_XCD_NT=1"
cd ../..
B=$PWD/eonerf_code_amd/csrc/build
run() { # label lib xcd
  EONERF_LIB=$2 EONERF_PIPE_XCD=$3 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload full 2> /dev/null | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.readline()); k = d['kernels']
    print('$1: full %.3f ms | pipe_cam %.4f pipe_sun %.4f wgrad %.4f' % (d['ms_per_step'], k['bwd_pipe_camera']['avg_ms'], k['bwd_pipe_sun']['avg_ms'], k['wgrad_gemm']['avg_ms']))
except Exception as e:
    print('$1: failed', e)"
}
for i in 1 2; do
  run "base (mixed pipelines, ring 16)      " "" 0
  run "xcd, plain stores + sc1 loads        " "" 1
  run "xcd, sc1 stores + sc1 loads (roles)  " $B/libeonerf_vxplain0.so 1
  run "xcd, plain stores + nt loads         " $B/libeonerf_vxnt.so 1
  run "xcd, plain + sc1, ring 8             " $B/libeonerf_vxring8.so 1
  run "xcd, plain + nt, ring 8              " $B/libeonerf_vxring8nt.so 1
  run "xcd, plain + nt, ring 4              " $B/libeonerf_vxring4nt.so 1
  run "mixed pipelines, ring 8              " $B/libeonerf_vxring8.so 0
done
