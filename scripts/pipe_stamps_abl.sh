#!/bin/bash
# Per-phase cycle stamps of the pipeline stages (scripts/pipe_stamps.py) for the shipped stage and for ablated ones (EO_PABL): where does the
# step's time go when the dW phase (128) or the operand loads (8) are taken away?
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o')
for N in 0 "$@"; do
  $HC -DEO_PIPE_STAMPS=1 -DEO_PABL=$N -c eonerf_bwd_pipe.hip -o build/pipe_vs$N.o && $HC -shared -o build/libeonerf_vs$N.so $REST build/pipe_vs$N.o || exit 1
done
cd ../..
for N in 0 "$@"; do
  echo "== EO_PABL=$N (camera pass of the rgb step: 7 stages x 8 waves, cycles per 32-sample step)"
  EONERF_LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_vs$N.so timeout -k 10 120 python scripts/pipe_stamps.py 2>&1 | grep -E "pipelines|stage|L7 \| w[04]|L6 \| w[04]|L4 \| w[04]|L1 \| w[04]"
done
