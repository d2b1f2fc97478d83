#!/bin/bash
# Diagnostic builds of the pipelined backward with parts of a step removed (EO_PABL bits, eonerf_bwd_pipe.hip): which part of a step
# costs what.  Usage (on the GPU box): bash scripts/pipe_ablate.sh 1 2 3 4 8 16 32 63   -> one line of per-step cycles per variant
cd $(dirname $0)/../eonerf_code_amd/csrc
OBJS=$(ls build/*.o | grep -v eonerf_bwd_pipe.o | grep -v pipe_abl)
for N in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DEO_PABL=$N -c eonerf_bwd_pipe.hip -o build/pipe_abl$N.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/libeonerf_abl$N.so $OBJS build/pipe_abl$N.o || exit 1
done
cd ../..
for N in 0 "$@"; do
  LIB=""; [ "$N" != "0" ] && LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_abl$N.so
  echo "== EO_PABL=$N"
  EONERF_LIB=$LIB timeout -k 10 120 python scripts/pipe_stamps.py 2>&1 | grep -E " 1 L6 \| w0| 1 L6 \| w4| 3 L4 \| w0| 3 L4 \| w4"
done
