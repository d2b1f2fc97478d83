# Do the intra-XCD hand-offs of EONERF_PIPE_XCD=1 come out of the XCD's L2?  FETCH_SIZE / TCC hit counters of k_bwd_pipe, both layouts.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; rm -rf $O; mkdir -p $O; cd $R
for X in 0 1; do
  export EONERF_PIPE_XCD=$X
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$X -o prof -- python3 bench.py --steps 4 --warmup 2 --workload full --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/fetch_$X.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$X -o prof -- python3 bench.py --steps 4 --warmup 2 --workload full --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/write_$X.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/hit_$X -o prof -- python3 bench.py --steps 4 --warmup 2 --workload full --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/hit_$X.err
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6i"
for d in sorted(glob.glob(O + "/*_[01]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_bwd_pipe" in r["Kernel_Name"] or "k_wgrad" in r["Kernel_Name"] or "k_mlp_fwd" in r["Kernel_Name"]:
                k = (r["Kernel_Name"][:60], r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (v, n) in sorted(acc.items()):
            print(os.path.basename(d), kn, cn, "avg per launch %.4g" % (v / n), "launches", n)
PY
