# sweep of the weight-gradient GEMM's work-item count (EONERF_WGRAD_ITEMS, 0 = default) on one box: kernel pass of one workload
for it in ${ITEMS:-0 390 520 650 780 1040 1560}; do
  EONERF_WGRAD_ITEMS=$it python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${WL:-full} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('items $it', round(d['ms_per_step'],3), 'ms/step; wgrad', round(d['kernels']['wgrad_gemm']['avg_ms'],3))
"
done
