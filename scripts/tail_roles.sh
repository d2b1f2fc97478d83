# which role of k_step_tail is its long pole: diagnostic builds (ab_libs/libeonerf_tailskip<bits>.so = csrc/eonerf_rays_bwd.hip compiled with
# -DEO_TAIL_SKIP=<bits> and linked with the other objects of csrc/build; bits: 1 bottleneck rows, 2 head rows, 4 embedding gradient are
# dropped; results wrong) against the in-tree library, kernel time from rocprofv3's kernel stats.  VARIANTS="base 5 6 3" bash scripts/tail_roles.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in ${VARIANTS:-base}; do
  if [ $v = base ]; then unset EONERF_LIB; else export EONERF_LIB=$R/ab_libs/libeonerf_tailskip$v.so; fi
  O=$R/gpurun_out/tail_roles/$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 10 --warmup 3 --workload full --no-cpu-baseline --no-kernel-pass > $O/bench.json 2> $O/err.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/t_kernel_stats.csv")):
    if "k_step_tail" in r["Name"]:
        print("skip=$v k_step_tail avg us", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2), "calls", r["Calls"])
PY
done
