"""Turns gpurun_out/prof_r02 (scripts/run_profiles.sh on the MI355X box) into the committed summaries under profiles/:
kernel stats per workload, HBM/fabric bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md prescribes, KB units; the first two launches of every kernel dropped), MFMA busy fraction and effective clock
from SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, and profiles/traffic.json (read by bench.py)."""
import collections
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.environ.get("RT", "r05")
SRC = os.path.join(REPO, "gpurun_out", "prof_" + RT)
DST = os.path.join(REPO, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else RT + "_a"
SHORT = {"k_mlp_fwd<PBf16, true, 2": "fwd_chain_camera", "k_mlp_fwd<PBf16, true, 1": "fwd_chain_camera", "k_mlp_fwd<PBf16, false, 1": "fwd_chain_sun",
         "k_mlp_bwd<PBf16, true, false, false,": "bwd_chain_camera", "k_mlp_bwd<PBf16, true, false, true,": "bwd_chain_camera",
         "k_mlp_bwd<PBf16, false, true, false,": "bwd_chain_sun", "k_bwd_pipe": "bwd_pipe", "k_wgrad<PBf16>": "wgrad_gemm", "k_ig_tail": "ig_tail_sun", "k_enc_pair": "ig_tail_sun"}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def counters(path, names):
    rows = list(csv.DictReader(open(path)))
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if r["Counter_Name"] in names:
            by[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return by


def durations(path):
    rows = list(csv.DictReader(open(path)))
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
    return d


traffic = {}
for wl in ("rgb", "full"):
    shutil.copy(os.path.join(SRC, f"stats_{wl}", "prof_kernel_stats.csv"), os.path.join(DST, f"{TAG}_bench_{wl}_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, f"bench_under_rocprof_{wl}.json"), os.path.join(DST, f"{TAG}_bench_{wl}_under_rocprof.json"))
    f = counters(os.path.join(SRC, f"pmc_fetch_{wl}", "prof_counter_collection.csv"), {"FETCH_SIZE"})
    w = counters(os.path.join(SRC, f"pmc_write_{wl}", "prof_counter_collection.csv"), {"WRITE_SIZE"})
    out = [["kernel", "launches_used", "fetch_bytes_per_launch(2xFETCH_SIZE)", "write_bytes_per_launch", "total_bytes_per_launch"]]
    tot = collections.defaultdict(float)
    step_total = 0.0
    # optimisation steps of the profiled run = launches of the Adam kernel (the bench's declared conditioning phase adds steps to the
    # `--steps 4 --warmup 2` of the command line)
    n_steps = float(max([len(v.get("FETCH_SIZE", [])) for k, v in f.items() if "k_adam" in k] + [1]))
    for name in sorted(set(f) | set(w)):
        fv = [v for _, v in sorted(f.get(name, {}).get("FETCH_SIZE", []))]
        wv = [v for _, v in sorted(w.get(name, {}).get("WRITE_SIZE", []))]
        # per launch; kernels launched more than once per step (pipe: sun + camera) are summed per step below
        fa = sum(fv[len(fv) // 3:]) / max(1, len(fv[len(fv) // 3:])) * 2 * 1024 if fv else 0.0
        wa = sum(wv[len(wv) // 3:]) / max(1, len(wv[len(wv) // 3:])) * 1024 if wv else 0.0
        if fa + wa < 1e6:
            continue
        per_step = (len(fv) if fv else len(wv)) / n_steps          # launches of this kernel per optimisation step
        out.append([name, len(fv) - len(fv) // 3, f"{fa:.0f}", f"{wa:.0f}", f"{fa + wa:.0f}"])
        s = short(name)
        if s:
            tot[s] += (fa + wa) * max(1.0, round(per_step))
        step_total += (fa + wa) * max(1.0, round(per_step))
    out.append(["ALL KERNELS, bytes per optimisation step", "", "", "", f"{step_total:.0f}"])
    with open(os.path.join(DST, f"{TAG}_pmc_hbm_traffic_{wl}.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(out)
    traffic[f"{wl}_bf16"] = {k: v for k, v in tot.items()}
    # k_bwd_pipe runs twice per step in the full workload (sun pass first, then the camera pass): split its dispatches by order
    pipe = [n for n in (set(f) | set(w)) if "k_bwd_pipe" in n]
    if pipe:
        fv = [v for _, v in sorted(f.get(pipe[0], {}).get("FETCH_SIZE", []))]
        wv = [v for _, v in sorted(w.get(pipe[0], {}).get("WRITE_SIZE", []))]
        if wl == "full":
            fs, fc, ws_, wc = fv[0::2][2:], fv[1::2][2:], wv[0::2][2:], wv[1::2][2:]
            avg = lambda x: sum(x) / max(1, len(x))
            traffic[f"{wl}_bf16"]["bwd_pipe_sun"] = avg(fs) * 2 * 1024 + avg(ws_) * 1024
            traffic[f"{wl}_bf16"]["bwd_pipe_camera"] = avg(fc) * 2 * 1024 + avg(wc) * 1024
        else:
            traffic[f"{wl}_bf16"]["bwd_pipe_camera"] = tot.get("bwd_pipe", 0.0)
    traffic[f"{wl}_bf16"]["step_total"] = step_total
    # MFMA busy
    m = counters(os.path.join(SRC, f"pmc_mfma_{wl}", "prof_counter_collection.csv"), {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
    dur = durations(os.path.join(SRC, f"pmc_mfma_{wl}", "prof_kernel_trace.csv"))
    out = [["kernel", "avg_ms", "mfma_busy_frac_of_simd_cycles", "effective_clock_GHz"]]
    for name in sorted(m):
        if not short(name):
            continue
        busy = [v for _, v in sorted(m[name]["SQ_VALU_MFMA_BUSY_CYCLES"])][2:]
        act = [v for _, v in sorted(m[name]["GRBM_GUI_ACTIVE"])][2:]
        ms = [v for _, v in sorted(dur[name])][2:]
        if not busy or not act:
            continue
        b, a_, t = sum(busy) / len(busy), sum(act) / len(act), sum(ms) / len(ms)
        out.append([name, f"{t:.4f}", f"{b / (a_ / 8 * 1024):.4f}", f"{a_ / 8 / (t * 1e-3) / 1e9:.3f}"])
    with open(os.path.join(DST, f"{TAG}_pmc_mfma_busy_{wl}.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(out)
for name in ("bench.json", "bench_chain_gemm_path.json", "bench_fp32.json"):
    shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{TAG}_{name}"))
traffic["_note"] = ("bytes per launch (per step for kernels launched twice a step) = (2*FETCH_SIZE + WRITE_SIZE)*1024: the L2's fabric-side "
                    "request counters (gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section; Infinity-Cache "
                    "hits are counted, so the ring hand-offs of the pipelined backward appear here although they never reach HBM); separate "
                    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `bench.py --steps 4 --warmup 2 --workload <wl>`; " + TAG)
# which build the counters describe: the commit the profiled snapshot was taken from (the profile run is started from a clean tree)
import subprocess
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(DST), capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "eonerf_code_amd", "bench.py"], cwd=os.path.dirname(DST), capture_output=True, text=True).stdout.strip())
except OSError:
    head, dirty = "unknown", False
traffic["_source"] = {"profile_set": TAG, "commit": head + ("+uncommitted" if dirty else ""), "files": [f"profiles/{TAG}_pmc_hbm_traffic_full.csv", f"profiles/{TAG}_pmc_hbm_traffic_rgb.csv"]}
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
