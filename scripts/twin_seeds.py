"""Twin training over several jitter keys: mean DSM MAE per precision with its spread (scripts/twin_training.py has the pairwise view).
Usage: python scripts/twin_seeds.py [n_seeds=6] [steps=2000] > gpurun_out/<dir>/twin_seeds.json"""
import json, os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from bf16_common import twin_train, export_quality  # noqa: E402
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
res = {"bf16": [], "fp32": []}
alts = {"bf16": [], "fp32": []}
for seed in range(7, 7 + n_seeds):
    for prec in ("bf16", "fp32"):
        t0 = time.time()
        q = export_quality(twin_train(prec, steps=steps, steps_per_epoch=steps // 4, noise_seed=seed))
        res[prec].append((100 * q["dsm_mae_m"], q["psnr"]))
        alts[prec].append(q["alt"])
        print(f"[twin] jitter {seed} {prec}: DSM MAE {100 * q['dsm_mae_m']:.2f} cm, PSNR {q['psnr']:.2f} dB ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
out = {"steps": steps, "n_seeds": n_seeds, "held_out_rays": int(alts["bf16"][0].numel())}
for prec in res:
    m = [r[0] for r in res[prec]]
    out[prec] = {"dsm_mae_cm": m, "mean": statistics.mean(m), "stdev": statistics.stdev(m), "psnr_mean": statistics.mean(r[1] for r in res[prec])}
d = out["bf16"]["mean"] - out["fp32"]["mean"]
se = (out["bf16"]["stdev"] ** 2 / n_seeds + out["fp32"]["stdev"] ** 2 / n_seeds) ** 0.5
out["mean_diff_bf16_minus_fp32_cm"] = d
out["standard_error_cm"] = se
# the ensemble-mean surfaces: per-ray altitude averaged over the seeds, bf16 minus fp32 (trajectory noise averages out, a systematic shift would not)
mb, mf = torch.stack(alts["bf16"]).mean(0), torch.stack(alts["fp32"]).mean(0)
out["ensemble_mean_surface_diff_cm"] = {"mean": 100 * (mb - mf).mean().item(), "mean_abs": 100 * (mb - mf).abs().mean().item(), "p99_abs": 100 * (mb - mf).abs().quantile(0.99).item()}
print(json.dumps(out, indent=1))
