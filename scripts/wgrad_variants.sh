# riders (sigma row / embedding columns on the bottleneck-factor job) on / off / the previous build of the library, one box, both workloads
for wl in full rgb; do for rd in 1 0 old 1 0 old; do
  if [ $rd = old ]; then export EONERF_LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_old.so; unset EONERF_WGRAD_RIDERS; else unset EONERF_LIB; export EONERF_WGRAD_RIDERS=$rd; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl riders $rd:', round(d['ms_per_step'],3), 'ms/step; wgrad', round(d['kernels']['wgrad_gemm']['avg_ms'],3))
"
done; done
