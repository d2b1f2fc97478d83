# Where the weight-gradient GEMM's time goes: diagnostic builds of eonerf_wgrad.hip with parts of its loop removed (EO_WG_ABL, results wrong).
#   build (in the container):  bash scripts/wgrad_ablate.sh build      -> ab_libs/libeonerf_wgabl<N>.so
#   run (on the GPU box):      bash scripts/wgrad_ablate.sh
if [ "$1" = build ]; then
  cd $(dirname $0)/../eonerf_code_amd/csrc && make -j8 > /dev/null && mkdir -p ../../ab_libs
  for N in 1 2 4 8 3 5 6 7; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEO_WG_ABL=$N -c eonerf_wgrad.hip -o /tmp/wgabl$N.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_libs/libeonerf_wgabl$N.so $(ls build/*.o | grep -v eonerf_wgrad.o) /tmp/wgabl$N.o
  done
  ls -la ../../ab_libs/; exit 0
fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wgrad_ablate; mkdir -p $O; cd $R
for N in 0 1 2 4 8 3 5 6 7 0; do
  L=$R/eonerf_code_amd/csrc/libeonerf_hip.so; [ $N != 0 ] && L=$R/ab_libs/libeonerf_wgabl$N.so
  for WL in full rgb; do
  EONERF_LIB=$L EONERF_BENCH_CONDITION=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $WL 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('abl $N (1 no flush, 2 no mfma, 4 no dma, 8 no barrier) $WL: wgrad %.4f ms' % d['kernels']['wgrad_gemm']['avg_ms'])
"
  done
done | tee $O/ablate.txt
