#!/bin/bash
# Diagnostic: builds libeonerf_stamp.so -- EO_STAMP: s_memtime stamps around the chain kernels' waits (scripts/stamp_probe.py);
# EO_PIPE_STAMPS: per-phase cycle sums of the pipelined backward's stages (scripts/pipe_stamps.py).  Run the scripts with EONERF_LIB
# pointing at it; the production library carries neither.
set -e
cd "$(dirname "$0")/../eonerf_code_amd/csrc"
make -j8 >/dev/null
mkdir -p build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DEO_STAMP -DEO_PIPE_STAMPS=1"
( /opt/rocm/bin/hipcc $FLAGS -c eonerf_mlp_fwd.hip -o build/abl/fwd_stamp.o & /opt/rocm/bin/hipcc $FLAGS -c eonerf_mlp_bwd.hip -o build/abl/bwd_stamp.o &
  /opt/rocm/bin/hipcc $FLAGS -c eonerf_bwd_pipe.hip -o build/abl/pipe_stamp.o & wait )
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/abl/libeonerf_stamp.so build/abl/fwd_stamp.o build/abl/bwd_stamp.o build/abl/pipe_stamp.o \
   build/eonerf_api.o build/eonerf_pack.o build/eonerf_rays.o build/eonerf_ig_tail.o build/eonerf_wgrad.o build/eonerf_rays_bwd.o build/eonerf_raygen.o
echo built build/abl/libeonerf_stamp.so
