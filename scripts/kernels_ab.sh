# per-kernel event times of two builds on the same box
for lib in old new; do
  if [ $lib = old ]; then export EONERF_LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_old.so; else unset EONERF_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${WL:-full} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['ms_per_step'],3), 'ms/step; kernels:', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, 'sum', round(sum(v['avg_ms'] for v in d['kernels'].values()),3))
"
done
