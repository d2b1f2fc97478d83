for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/atomics $i /"
  EONERF_PIPE_PARTIALS=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/partials $i /"
done
