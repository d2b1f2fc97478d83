#!/bin/bash
# eonerf_enc_pair.hip with parts removed (EO_EP_ABL: 1 flush, 2 d enc waves, 4 dW waves, 8 LDS-DMA refill): where does its launch go?
cd $(dirname $0)/../eonerf_code_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
REST=$(ls build/*.o | grep -v -E 'eonerf_enc_pair.o|_v[0-9a-z]*\.o')
for N in "$@"; do $HC -DEO_EP_ABL=$N -c eonerf_enc_pair.hip -o build/ep_v$N.o && $HC -shared -o build/libeonerf_vep$N.so $REST build/ep_v$N.o || exit 1; done
cd ../..
for N in 0 "$@"; do
  LIB=""; [ "$N" != "0" ] && LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_vep$N.so
  EONERF_LIB=$LIB timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload full 2> /dev/null | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.readline()); k = d['kernels']
    print('EO_EP_ABL=$N: enc_pair %.4f ms | wgrad %.4f | step %.3f' % (k['ig_tail_sun']['avg_ms'], k['wgrad_gemm']['avg_ms'], d['ms_per_step']))
except Exception as e:
    print('EO_EP_ABL=$N: failed', e)"
done
