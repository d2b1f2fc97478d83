# same-box A/B of two builds of the library with the same ABI: EONERF_LIB=ab_libs/libeonerf_${A:-A}.so against the in-tree one,
# alternating runs (N=${N:-3}).  Old builds are made from a git worktree of the commit to compare with and copied to ab_libs/.
for i in $(seq 1 ${N:-3}); do
  EONERF_LIB=$PWD/ab_libs/libeonerf_${A:-A}.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/old $i /"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/new $i /"
done
