# sweep of the weight-gradient GEMM's work-item count (EONERF_WGRAD_ITEMS; default: ~4 items per CU)
for N in 0 256 448 504 512 640 768 1024 1536 2048; do
  EONERF_WGRAD_ITEMS=$N python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('items $N: rgb %.3f ms (gemm %.4f)  full %.3f ms (gemm %.4f)' % (d['ms_per_step'], d['kernels']['wgrad_gemm']['avg_ms'], d['full']['ms_per_step'], d['full']['kernels']['wgrad_gemm']['avg_ms']))"
done
