"""GPU: what the benchmarked precision (bf16 MFMA) does to the outputs at the BENCH size -- 4096 rays x 128 samples, ~4.9e5 samples
per pass -- measured against the fp32-HIP path on identical rays, jitter and weights.  The fp32 path is itself pinned to the
reference goldens / the oracle at 1e-4 (tests/test_hip_forward.py, test_hip_backward.py), so this chains bf16 to the reference at
full size, where the CPU oracle is too slow to be the checker.

Two weight states (tests/bf16_common.py):
  (a) Xavier-uniform init (mlp.py:22-28): smooth, flat density;
  (b) the same field after 400 optimisation steps (autograd path, in-kernel jitter) on a synthetic terrain with depth + colour
      supervision: density concentrated around a surface -- the regime a DSM is exported from.
Bounds (measured on MI355X, round 2: scripts/bf16_vs_fp32.py; values in DESIGN.md 4).  State (b) depends on where the training run
ends; seven runs of different builds / atomics orderings gave the ranges quoted below, the bounds are ~2x their worst case, and the
test trains with fixed-order sums so that one build always reproduces one trajectory:
  rgb            max |d| <= 5e-3 (a) / 5e-2 (b: 0.007-0.028),  mean <= 1e-4 / 1e-3 (b: 3.5e-4-4.8e-4)
  depth          mean |d| <= 1.5e-4 (a) / 6e-4 (b: 1.7e-4-3.2e-4)   (normalised units)
  altitude       datasets/satellite.py:502-533 at Z_scale = 50 m, per ray |alt_bf16 - alt_fp32|: mean <= 0.5 cm (a) / 3 cm (b: 0.8-1.6 cm),
                 99th percentile <= 2 cm (a) / 10 cm (b: 3.4-5.2 cm)
  DSM MAE        the north-star criterion ("DSM altitude MAE within 1 cm of reference"): the altitude MAE against the synthetic
                 terrain, computed for each path, differs by <= 1 cm -- measured 0.1 cm (the per-ray differences are nearly
                 zero-mean and tiny next to the metre-scale error of a partially trained field)
  gradients      one full train step (shadow pass + uncertainty loss), per tensor: cosine >= 0.999 (a) / 0.98 (b) against the fp32
                 path's gradient, relative L2 error <= 3e-2 (a) / 2.5e-1 (b).  The worst tensor is always layer 0's weight (the
                 product against the 2^9-frequency encodings); in (b) its numbers depend on where the (atomics-ordered, hence not
                 bit-reproducible) training run ended: cosine 0.9905-0.9973, relative error 0.073-0.157 over six trajectories
The per-sample noise of bf16 activations averages out along a ray; what remains in (b) is mostly the systematic part (bf16-rounded
WEIGHTS shift the learned surface by a fraction of a sample spacing).  A field trained in bf16 mode has learned with those rounded
weights; for a bit-faithful export the same checkpoint renders in precision="fp32" (1e-4 parity mode).
"""
import json

import pytest
import torch

from bf16_common import compare_precisions, make_fields, train_on_terrain

pytestmark = pytest.mark.gpu


def _check(st, rgb_max, rgb_mean, depth_mean, alt_mae, alt_p99, cos_min, rel_max):
    print("bf16 vs fp32 at 4096 x 128:", json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in st.items()}))
    assert st["n_samples_equal"], "the sampler is precision independent: sample counts must be identical"
    assert st["rgb_max"] <= rgb_max and st["rgb_mean"] <= rgb_mean, (st["rgb_max"], st["rgb_mean"])
    assert st["depth_mean"] <= depth_mean, st["depth_mean"]
    assert st["alt_mae_m"] <= alt_mae, f"altitude MAE {st['alt_mae_m'] * 100:.2f} cm"
    assert st["alt_p99_m"] <= alt_p99, f"altitude p99 {st['alt_p99_m'] * 100:.2f} cm"
    assert abs(st["dsm_mae_bf16_m"] - st["dsm_mae_fp32_m"]) <= 0.01, (st["dsm_mae_bf16_m"], st["dsm_mae_fp32_m"])
    assert st["grad_cos_min"] >= cos_min, (st["grad_cos_min_name"], st["grad_cos_min"])
    assert st["grad_rel_max"] <= rel_max, st["grad_rel_max"]


def test_bf16_vs_fp32_xavier_init_full_size():
    f16, f32 = make_fields(seed=42)
    _check(compare_precisions(f16, f32, seed=1), rgb_max=5e-3, rgb_mean=1e-4, depth_mean=1.5e-4, alt_mae=0.005, alt_p99=0.02,
           cos_min=0.999, rel_max=3e-2)


def test_bf16_vs_fp32_trained_field_dsm_mae_within_1cm_full_size(monkeypatch):
    # fixed-order gradient sums (EONERF_DETERMINISTIC, read when the field creates its native context): the 400-step training run then
    # ends in the same state every time this build runs, instead of wherever the fp32 atomics' ordering takes it
    monkeypatch.setenv("EONERF_DETERMINISTIC", "1")
    f16, f32 = make_fields(seed=42)
    train_on_terrain(f16, 400)
    f32.load_state_dict(f16.state_dict())
    st = compare_precisions(f16, f32, seed=1)
    assert st["depth_err_vs_terrain_mean"] < 0.1                    # the field did learn the terrain (from 0.19 at init)
    _check(st, rgb_max=5e-2, rgb_mean=1e-3, depth_mean=6e-4, alt_mae=0.030, alt_p99=0.10, cos_min=0.98, rel_max=2.5e-1)


@pytest.mark.parametrize("eval_precision", ["fp16x3", "fp32"])
def test_export_render_of_a_bf16_trained_field_runs_in_fp32_and_matches_the_oracle(monkeypatch, eval_precision):
    """The fence around the bf16 altitude shift: EXPORT renders -- render_image(eval=True), eval_eonerf.py:311-324, or a
    module in .eval() mode under no_grad, train_eonerf.py:197-226 -- of a field TRAINED in bf16 run on the module's second context
    (EONerfMLP.eval_precision: "fp16x3", the default since round 4 -- hi + lo fp16 operands, three fp16 MFMAs per product -- or "fp32",
    exact fp32 FMA chains) and therefore match the reference arithmetic on the same checkpoint: every output within 1e-4 of the oracle
    at 4096 x 128, altitude within 1 cm (Z_scale 50 m) on EVERY ray -- the same bar for both."""
    from oracle import eonerf_oracle as orc
    from bf16_common import R, N_IMG, STEP, Z_SCALE, terrain_batch
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    monkeypatch.setenv("EONERF_DETERMINISTIC", "1")
    f16, _ = make_fields(seed=42)
    assert f16.eval_precision == "fp16x3"                              # the default
    f16.eval_precision = eval_precision                                # (read when the export context is created, at the first export)
    train_on_terrain(f16, 400)
    sd = {k: v.detach().cpu() for k, v in f16.state_dict().items()}
    rays, _, _, _ = terrain_batch(R, seed=903)
    img = torch.zeros(R, dtype=torch.int64, device="cuda")            # eval_eonerf.py:300-304: ts all zeros
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
    sr = define_satrays_from_tensors(rays, img[:, None])
    with torch.no_grad():
        exp, n_exp = render_image(f16, None, sr, None, None, epoch_idx=3, chunk=R, render_step_size=STEP, noise=[noise], eval=True)
        f16.eval()                                                     # the validation loop's way into the same path
        val, n_val = render_image(f16, None, sr, None, None, epoch_idx=3, chunk=R, render_step_size=STEP, noise=[noise])
        f16.train()
        raw, _ = render_image(f16, None, sr, None, None, epoch_idx=3, chunk=R, render_step_size=STEP, noise=[noise])   # training-mode render: bf16
        outs = []
        for i in range(0, R, 512):                                     # the oracle, ray chunk by ray chunk (memory)
            sl = slice(i, i + 512)
            o, _ = orc.render_rays(orc.Field(sd), orc.define_satrays_from_tensors(rays[sl].cpu(), img[sl, None].cpu()),
                                   noise[0][sl].cpu(), noise[2][sl].cpu(), 3, STEP, eval=True)
            outs.append(o)
    ref = torch.cat(outs)
    assert f16._ctx_eval is not None                                   # the export context exists and was used
    worst = {}
    for name, (a, b) in orc.RESULT_SLICES.items():
        got, want = exp[name].cpu(), ref[:, a:b]
        if name in ("pts_per_ray", "sc_pts_per_ray", "entropy", "opacity_after_surface"):
            assert torch.equal(got, want), name
        else:
            worst[name] = (got - want).abs().max().item()
            assert worst[name] < 1e-4, (name, worst[name])
        if name not in ("rgb", "shadowless_rgb"):                      # .eval() mode without eval=True: per-ray radiometric rows (all row 0 here)
            assert torch.equal(val[name], exp[name]), name
    alt = orc.altitude_from_depth(rays.cpu(), exp["depth"].cpu(), Z_SCALE, 20.0)
    alt_ref = orc.altitude_from_depth(rays.cpu(), ref[:, 3:4], Z_SCALE, 20.0)
    assert (alt - alt_ref).abs().max().item() < 0.01                   # 1 cm, every ray
    d_bf16 = (orc.altitude_from_depth(rays.cpu(), raw["depth"].cpu(), Z_SCALE, 20.0) - alt_ref).abs()
    print(f"export ({eval_precision} context) max output errors " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()) +
          f"; max altitude error {(alt - alt_ref).abs().max().item() * 100:.4f} cm; bf16 render of the same weights: "
          f"mean {d_bf16.mean().item() * 100:.2f} cm, p99 {d_bf16.quantile(0.99).item() * 100:.2f} cm")
    assert d_bf16.mean().item() > 1e-4                                 # (the two paths really are different kernels)


def test_twin_training_bf16_vs_fp32_dsm_mae():
    """VERDICT r5 #3 -- north_star: "DSM MAE matching the reference within 1 cm", for a field TRAINED in the throughput mode.  The same
    scene is trained through FusedTrainer, 2,000 steps across the loss switch (epochs 0-3: MSE, then shadow pass + uncertainty loss; StepLR;
    depth prior on every fourth ray as train_eonerf.py:145-149) -- precision="bf16" (the 1.25 M rays/s mode) and precision="fp32" (the
    1e-4 parity mode = the reference's arithmetic) with identical initial weights, ray table and batch order, under THREE jitter keys each.
    All fields are exported in their default export precision on 16,384 held-out rays; the statistic is the DSM altitude MAE against the
    terrain (what train_eonerf.py:194-294 / sat_utils.py:226 measure against the lidar DSM), Z_scale 50 m.

    What round 6 measured (scripts/twin_seeds.py, profiles/r06_twin_training_seeds.json): after 2,000 steps every run sits at a
    DSM MAE of 1.9-2.1 m; over six keys bf16 196.12 +- 3.7 cm, fp32 196.07 +- 7.3 cm (fp32 has a tail: 189 ... 210), two runs of the SAME
    arithmetic differ by 2-20 cm -- training is chaotic, atomics order and jitter are enough -- and bf16 against fp32 under one key by 1-13 cm.
    A 1-cm difference of MAEs is not resolvable by a pair of trainings, and neither is a 10-cm one (the first form of this test, one pair
    with a 10-cm bound, failed on its third run with fp32 = 204.7, fp32 under another key = 194.5, bf16 = 192.0).  What the data supports
    and this test asserts, on the MEANS of three runs per precision (sigma of their difference: 4.7 cm): the precision change moves the
    mean DSM MAE by less than 15 cm (7 % of it), no bf16 run lies outside the fp32 runs' range widened by 15 cm, the mean surface does not
    shift (|mean per-ray difference of the ensemble-mean altitudes| <= 10 cm; measured -1.5 cm over six keys, +4.1 cm over three), PSNR within 1 dB.  The line it prints has every run."""
    from bf16_common import twin_train, export_quality
    keys = (7, 8, 9)
    q = {p: [export_quality(twin_train(p, noise_seed=k)) for k in keys] for p in ("bf16", "fp32")}
    mae = {p: [v["dsm_mae_m"] for v in q[p]] for p in q}
    mean = {p: sum(mae[p]) / len(keys) for p in q}
    psnr = {p: sum(v["psnr"] for v in q[p]) / len(keys) for p in q}
    surf = {p: torch.stack([v["alt"] for v in q[p]]).mean(0) for p in q}
    d = surf["bf16"] - surf["fp32"]
    ad = d.abs()
    diff = abs(mean["bf16"] - mean["fp32"])
    rep = {"dsm_mae_bf16_cm": [100 * v for v in mae["bf16"]], "dsm_mae_fp32_cm": [100 * v for v in mae["fp32"]],
           "mean_bf16_cm": 100 * mean["bf16"], "mean_fp32_cm": 100 * mean["fp32"], "mean_diff_cm": 100 * diff, "within_1cm": diff <= 0.01,
           "psnr_bf16": psnr["bf16"], "psnr_fp32": psnr["fp32"],
           "ensemble_mean_surface_diff_cm": {"mean": 100 * d.mean().item(), "mean_abs": 100 * ad.mean().item(), "p50": 100 * ad.quantile(0.5).item(),
                                             "p90": 100 * ad.quantile(0.9).item(), "p99": 100 * ad.quantile(0.99).item(), "max": 100 * ad.max().item()}}
    print("twin training (2000 steps, 3 jitter keys per precision, export renders on 16384 held-out rays):", json.dumps(rep))
    assert all(m < 3.0 for p in mae for m in mae[p]), "the fields did not learn the terrain (9.5 m at initialisation)"
    assert diff <= 0.15, rep
    assert min(mae["fp32"]) - 0.15 <= min(mae["bf16"]) and max(mae["bf16"]) <= max(mae["fp32"]) + 0.15, rep
    assert abs(d.mean().item()) <= 0.10, rep
    assert abs(psnr["bf16"] - psnr["fp32"]) <= 1.0, rep
