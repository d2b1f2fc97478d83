# same-box A/B of two builds of the library: EONERF_LIB=<old build> vs the in-tree one, alternating runs (N=${N:-3})
for i in $(seq 1 ${N:-3}); do
  EONERF_LIB=$PWD/eonerf_code_amd/csrc/build/libeonerf_old.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/old $i /"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass 2>&1 | grep "^\[bench\]" | sed "s/^/new $i /"
done
