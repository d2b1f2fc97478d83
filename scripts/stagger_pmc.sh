# VERDICT r5 #2's counters for the wave-stagger A/B of the chain kernels: SQ_WAIT_ANY, SQ_VALU_MFMA_COEXEC_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES per
# launch with EONERF_STAGGER = 0 (off), 1 (waves 4..7 late), 2 (odd waves late); one --pmc pass per setting, kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stagger_pmc
rm -rf $O; mkdir -p $O
for S in 0 1 2; do
  export EONERF_STAGGER=$S
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/s$S -o s -- python3 $R/bench.py --steps 6 --warmup 2 --workload full --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/s$S.err || { tail -3 $O/s$S.err; exit 1; }
done
unset EONERF_STAGGER
python3 - <<PY
import collections, csv
names = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU")
print("kernel | stagger | launches | " + " | ".join(names) + " | wait_any/wave_cycles | coexec/mfma_busy")
for S in (0, 1, 2):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$O/s%d/s_counter_collection.csv" % S)):
        k = r["Kernel_Name"]
        if "k_mlp_fwd" in k or "k_mlp_bwd<PBf16, true" in k:
            by[k[:64]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k in sorted(by):
        v = {n: [x for _, x in sorted(by[k][n])][2:] for n in names}
        m = {n: sum(v[n]) / max(1, len(v[n])) for n in names}
        print(k, "|", S, "|", len(v[names[0]]), "|", " | ".join("%.4g" % m[n] for n in names), "| %.3f | %.3f" % (m["SQ_WAIT_ANY"] / max(1.0, m["SQ_WAVE_CYCLES"]), m["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(1.0, m["SQ_VALU_MFMA_BUSY_CYCLES"])))
PY
