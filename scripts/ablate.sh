#!/bin/bash
# Diagnostic: builds libeonerf_abl<N>.so variants (EO_ABL=N, see eonerf_common.h) of the two chain kernels next to the
# shipped library.  Usage: scripts/ablate.sh 1 2 4 ...   then   EONERF_LIB=<path> python scripts/ablate.py
set -e
cd "$(dirname "$0")/../eonerf_code_amd/csrc"
make -j8 >/dev/null
mkdir -p build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DEO_ABL=$n -c eonerf_mlp_fwd.hip -o build/abl/fwd$n.o &
    /opt/rocm/bin/hipcc $FLAGS -DEO_ABL=$n -c eonerf_mlp_bwd.hip -o build/abl/bwd$n.o & wait )
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/abl/libeonerf_abl$n.so build/abl/fwd$n.o build/abl/bwd$n.o \
     build/eonerf_api.o build/eonerf_pack.o build/eonerf_rays.o build/eonerf_wgrad.o build/eonerf_rays_bwd.o build/eonerf_raygen.o
  echo built build/abl/libeonerf_abl$n.so
done
