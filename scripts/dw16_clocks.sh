# the 16x16x32 MFMA shape where it needs no re-layout (the weight-gradient half of the pipelined stage, EO_PIPE_DW16=1)
# against the 32x32x16 build on ONE box, with the clock the chip holds under either: kernel time from the kernel trace, effective clock
# = GRBM_GUI_ACTIVE / 8 XCDs / kernel time, matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles).
# ab_libs/libeonerf_dw16.so: the same sources built with -DEO_PIPE_DW16=1 (git worktree + make).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dw16; rm -rf $O; mkdir -p $O; cd $R
for V in base dw16 base dw16; do
  L=$R/eonerf_code_amd/csrc/libeonerf_hip.so; [ $V = dw16 ] && L=$R/ab_libs/libeonerf_dw16.so
  export EONERF_LIB=$L
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-pass --workload full 2>&1 | grep "^\[bench\]" | sed "s/^/$V /"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$V -o prof -- python3 bench.py --steps 6 --warmup 3 --workload full --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_$V.err
done
python3 - <<PY
import csv, collections
for V in ("base", "dw16"):
    rows = list(csv.DictReader(open("$O/pmc_%s/prof_counter_collection.csv" % V)))
    dur = {}
    for r in csv.DictReader(open("$O/pmc_%s/prof_kernel_trace.csv" % V)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "k_bwd_pipe" in r["Kernel_Name"] or "k_mlp_fwd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    for k, c in acc.items():
        g = c["GRBM_GUI_ACTIVE"][len(c["GRBM_GUI_ACTIVE"]) // 3:]
        m = c["SQ_VALU_MFMA_BUSY_CYCLES"][len(c["SQ_VALU_MFMA_BUSY_CYCLES"]) // 3:]
        t = sum(d for _, d in g) / len(g)
        clk = sum(v / 8 / d for v, d in g) / len(g)
        busy = sum(v / (4 * 256 * (gg / 8)) for (v, _), (gg, _) in zip(m, g)) / len(g)
        print(f"{V:5s} {k:60s} avg {t*1e3:.4f} ms  clock {clk/1e9:.3f} GHz  mfma busy {busy:.3f}")
PY
