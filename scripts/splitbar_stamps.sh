#!/bin/bash
# Per-phase cycle stamps of the pipeline stage with the workgroup barrier (EO_PIPE_SPLITBAR=0) and with arrival counters (1).
# Column "tr reads" = waves 0-3's wait for the free slot in front of their DMA issue (part of "DMA issue"); "barrier" = the start-of-step wait.
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_bwd_pipe.o|_v[0-9a-z]*\.o|pipe_abl')
for N in 0 1; do
  $HC -DEO_PIPE_STAMPS=1 -DEO_PIPE_SPLITBAR=$N $EXTRA -c eonerf_bwd_pipe.hip -o build/pipe_vt$N.o && $HC -shared -o build/libeonerf_vt$N.so $REST build/pipe_vt$N.o || exit 1
done
cd ../..
for N in 0 1; do
  echo "== EO_PIPE_SPLITBAR=$N (camera pass of the rgb step: 7 stages x 8 waves, cycles per 32-sample step)"
  EONERF_LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_vt$N.so timeout -k 10 120 python scripts/pipe_stamps.py 2>&1 | grep -E "pipelines|stage|L7 \| w[0145]|L6 \| w[04]|L4 \| w[0145]|L1 \| w[04]"
done
