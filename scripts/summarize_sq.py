"""gpurun_out/sq_<RT> (scripts/run_sq_counters.sh) -> profiles/<tag>_pmc_sq.csv (where the waves of the MFMA kernels spend their cycles) and
profiles/<tag>_pmc_lds.csv (LDS instructions, active cycles, bank-conflict cycles), per launch, first two launches of every kernel dropped."""
import collections
import csv
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.environ.get("RT", "r05")
SRC = os.path.join(REPO, "gpurun_out", "sq_" + RT)
TAG = sys.argv[1] if len(sys.argv) > 1 else RT + "_b"
KEEP = ("k_mlp_fwd", "k_mlp_bwd", "k_bwd_pipe", "k_wgrad", "k_ig_tail", "k_enc_pair")


def load(sub):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(SRC, sub, "s_counter_collection.csv"))):
        if any(k in r["Kernel_Name"] for k in KEEP):
            by[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: {c: (lambda v: sum(v) / max(1, len(v)))([x for _, x in sorted(vals)][2:]) for c, vals in d.items()} for k, d in by.items()}


a, b, c = load("a"), load("b"), load("c")
rows = [["kernel", "SQ_WAVE_CYCLES", "wait_any_frac", "wait_inst_any_frac", "active_inst_any_frac", "valu_frac", "lds_frac", "vmem_frac", "scalar_frac"]]
for k in sorted(a):
    w = a[k].get("SQ_WAVE_CYCLES", 0.0) or 1.0
    f = lambda d, n: f"{d.get(k, {}).get(n, 0.0) / w:.3f}"
    rows.append([k, f"{w:.0f}", f(a, "SQ_WAIT_ANY"), f(a, "SQ_WAIT_INST_ANY"), f(a, "SQ_ACTIVE_INST_ANY"), f(b, "SQ_ACTIVE_INST_VALU"), f(b, "SQ_ACTIVE_INST_LDS"),
                 f(b, "SQ_ACTIVE_INST_VMEM"), f(b, "SQ_ACTIVE_INST_SCA")])
csv.writer(open(os.path.join(REPO, "profiles", f"{TAG}_pmc_sq.csv"), "w", newline="")).writerows(rows)
rows = [["kernel", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "conflict_frac_of_active"]]
for k in sorted(c):
    d = c[k]
    act = d.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
    rows.append([k, f"{d.get('SQ_INSTS_LDS', 0):.0f}", f"{act:.0f}", f"{d.get('SQ_LDS_BANK_CONFLICT', 0):.0f}", f"{d.get('SQ_LDS_BANK_CONFLICT', 0) / act:.3f}"])
csv.writer(open(os.path.join(REPO, "profiles", f"{TAG}_pmc_lds.csv"), "w", newline="")).writerows(rows)
print(open(os.path.join(REPO, "profiles", f"{TAG}_pmc_sq.csv")).read())
print(open(os.path.join(REPO, "profiles", f"{TAG}_pmc_lds.csv")).read())
