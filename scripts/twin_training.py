"""Twin-training quality experiment, the long form of tests/test_bf16_fullsize.py::test_twin_training_...: the same scene
trained in bf16 mode and in fp32 mode, PLUS a second fp32 run under another jitter key -- the spread between two runs of the SAME arithmetic
is the yardstick for what the precision change does.  Usage (GPU box): python scripts/twin_training.py [steps] > gpurun_out/twin.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bf16_common import twin_train, export_quality  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
runs = {}
for tag, prec, seed in (("bf16", "bf16", 7), ("fp32", "fp32", 7), ("fp32_other_jitter", "fp32", 8), ("bf16_other_jitter", "bf16", 8)):
    t0 = time.time()
    q = export_quality(twin_train(prec, steps=steps, steps_per_epoch=steps // 4, noise_seed=seed))
    torch.cuda.synchronize()
    runs[tag] = q
    print(f"[twin] {tag}: DSM MAE {100 * q['dsm_mae_m']:.2f} cm, PSNR {q['psnr']:.2f} dB, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)


def diff(a, b):
    d = runs[a]["alt"] - runs[b]["alt"]
    ad = d.abs()
    return {"dsm_mae_diff_cm": 100 * (runs[a]["dsm_mae_m"] - runs[b]["dsm_mae_m"]), "psnr_diff_db": runs[a]["psnr"] - runs[b]["psnr"],
            "per_ray_alt_diff_cm": {"mean": 100 * d.mean().item(), "mean_abs": 100 * ad.mean().item(), "p50": 100 * ad.quantile(0.5).item(),
                                    "p90": 100 * ad.quantile(0.9).item(), "p99": 100 * ad.quantile(0.99).item(), "max": 100 * ad.max().item()}}


print(json.dumps({"steps": steps, "rays_per_step": 4096, "held_out_rays": int(runs["bf16"]["alt"].numel()), "z_scale_m": 50.0,
                  "runs": {k: {"dsm_mae_cm": 100 * v["dsm_mae_m"], "psnr_db": v["psnr"]} for k, v in runs.items()},
                  "bf16_vs_fp32_same_jitter": diff("bf16", "fp32"),
                  "fp32_vs_fp32_other_jitter": diff("fp32", "fp32_other_jitter"),
                  "bf16_vs_bf16_other_jitter": diff("bf16", "bf16_other_jitter")}, indent=1))
