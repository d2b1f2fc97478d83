"""Shared by tests/test_bf16_fullsize.py and scripts/bf16_vs_fp32.py: bf16-HIP against fp32-HIP at the bench size."""
import torch

R, N_IMG, STEP, Z_SCALE = 4096, 19, 2.0 / 128, 50.0


def make_fields(seed=42):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from oracle import eonerf_oracle as orc
    sd = orc.random_state_dict(N_IMG, seed=seed)                    # Xavier-uniform weights, zero biases (mlp.py:22-28)
    out = []
    for prec in ("bf16", "fp32"):
        f = EONerfMLP(N_IMG, radiometric_normalization=True, precision=prec)
        f.load_state_dict(sd, strict=True)
        out.append(f.cuda())
    return out


def terrain_height(x, y):
    return 0.25 * torch.sin(3.0 * x) * torch.cos(2.5 * y) - 0.2


def terrain_batch(n, seed):
    """JAX_068-like rays + the depth at which each ray meets a smooth synthetic terrain + a colour that depends on
    the hit point (what multi-view supervision would teach a real scene)."""
    from eonerf_code_amd.synthetic import synthetic_batch
    rays, img, _ = synthetic_batch(n, N_IMG, seed=seed)
    o, d = rays[:, :3], rays[:, 3:6]
    t = torch.full((n,), 1.0)
    for _ in range(30):                                            # fixed point of o_z + t d_z = h(x(t), y(t))
        p = o + d * t[:, None]
        t = (terrain_height(p[:, 0], p[:, 1]) - o[:, 2]) / d[:, 2]
    p = o + d * t[:, None]
    rgb = 0.5 + 0.4 * torch.stack([torch.sin(5 * p[:, 0]), torch.cos(4 * p[:, 1]), torch.sin(3 * p[:, 0] + 2 * p[:, 1])], 1)
    return rays.cuda(), img.cuda(), rgb.cuda(), t[:, None].cuda()


def train_on_terrain(f, steps, lr=5e-4):
    """autograd path (render_image) with depth + colour supervision, torch.optim.Adam; jitter from the in-kernel Philox stream"""
    import torch.nn.functional as F
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    opt = torch.optim.Adam(f.parameters(), lr=lr)
    f.set_noise_seed(7)
    for it in range(steps):
        rays, img, rgb, depth = terrain_batch(R, seed=100 + it % 16)
        res, _ = render_image(f, None, define_satrays_from_tensors(rays, img[:, None]), None, None, epoch_idx=0, chunk=R, render_step_size=STEP)
        loss = F.mse_loss(res["rgb"], rgb) + 10.0 * F.mse_loss(res["depth"], depth)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return float(loss)


def compare_precisions(f16, f32, seed=1, epoch=3):
    """-> dict of error statistics of the bf16 path against the fp32 path on ONE identical batch (same weights, rays, noise)."""
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors, get_utmalt_from_nerf_prediction
    from eonerf_code_amd.trainer import FusedTrainer
    rays, img, rgb, depth_gt = terrain_batch(R, seed=900 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
    sr = define_satrays_from_tensors(rays, img[:, None])
    outs = []
    with torch.no_grad():
        for f in (f16, f32):
            res, n = render_image(f, None, sr, None, None, epoch_idx=epoch, chunk=R, render_step_size=STEP, noise=[noise])
            outs.append((res, n))
    (a, na), (b, nb) = outs
    off, sc = [0.0, 0.0, 20.0], [250.0, 250.0, Z_SCALE]
    alt16 = get_utmalt_from_nerf_prediction(rays, a["depth"], off, sc)[2]
    alt32 = get_utmalt_from_nerf_prediction(rays, b["depth"], off, sc)[2]
    st = {"n_samples_equal": na == nb}
    for k in ("rgb", "depth", "albedo_rgb", "geo_shadows", "transient_s", "beta"):
        d = (a[k] - b[k]).abs()
        st[f"{k}_max"], st[f"{k}_mean"] = d.max().item(), d.mean().item()
    dalt = (alt16 - alt32).abs()
    st["alt_mae_m"], st["alt_max_m"], st["alt_p99_m"] = dalt.mean().item(), dalt.max().item(), dalt.quantile(0.99).item()
    st["depth_err_vs_terrain_mean"] = (b["depth"] - depth_gt).abs().mean().item()
    # the DSM-quality metric: altitude MAE against the (synthetic) ground truth, per precision -- north_star's criterion compares
    # THIS number between the two paths ("DSM altitude MAE within 1 cm of reference")
    alt_gt = get_utmalt_from_nerf_prediction(rays, depth_gt, off, sc)[2]
    st["dsm_mae_bf16_m"] = (alt16 - alt_gt).abs().mean().item()
    st["dsm_mae_fp32_m"] = (alt32 - alt_gt).abs().mean().item()
    # how concentrated sigma is: mean over rays of the weight-averaged |t - depth| is not exposed; use depth spread between two jitters
    # gradients of one full train step (uncertainty loss), same noise
    grads = []
    for f in (f16, f32):
        tr = FusedTrainer(f, lr=0.0, max_rays=R)
        before = f.flat_params().clone()
        tr.step(rays, img, rgb, epoch, noise=noise)
        grads.append(tr.d_flat.clone())
        f.flat_params().copy_(before)                               # lr = 0, but Adam's eps path: restore exactly
        f._packed_version = None
    cos, rel = {}, {}
    for (name, p), g16, g32 in zip(f32.named_parameters(), f16.grad_views(grads[0]), f32.grad_views(grads[1])):
        x, y = g16.flatten().double(), g32.flatten().double()
        if y.norm() == 0:
            continue
        cos[name] = (torch.dot(x, y) / (x.norm() * y.norm() + 1e-300)).item()
        rel[name] = ((x - y).norm() / y.norm()).item()
    st["grad_cos_min"] = min(cos.values())
    st["grad_cos_min_name"] = min(cos, key=cos.get)
    st["grad_rel_max"] = max(rel.values())
    st["grad_cos_trunk_min"] = min(v for k, v in cos.items() if k.startswith("base_mlp"))
    return st


def twin_train(precision, steps=2000, steps_per_epoch=500, table_batches=16, noise_seed=7, weight_seed=42, prior_every=4, w_depth0=10.0, w_decay=0.8):
    """ONE training run of the twin-training quality test: the launcher's loop (train_dp.py; train_eonerf.py:98-161,304-306)
    through FusedTrainer on the synthetic terrain -- colour supervision on every ray, a depth prior (metrics.depth_loss_L2 through the
    aux_loss hook, train_eonerf.py:145-149) on every fourth ray, MSE for epochs 0-1, shadow pass + uncertainty loss from epoch 2, StepLR 0.9
    and w_depth x 0.8 per epoch.  Everything but `precision` is identical between two calls: weights (seed), ray table, batch order, jitter
    key.  Returns the trained field."""
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer
    from oracle import eonerf_oracle as orc
    sd = orc.random_state_dict(N_IMG, seed=weight_seed)
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision=precision)
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    tr = FusedTrainer(f, lr=5e-4, max_rays=R, keep_message=False)
    tr.set_noise_seed(noise_seed)
    table = [terrain_batch(R, seed=100 + k) for k in range(table_batches)]
    prior_mask = (torch.arange(R, device="cuda") % prior_every == 0)
    lr, w_depth = 5e-4, w_depth0
    for it in range(steps):
        epoch = it // steps_per_epoch
        if it > 0 and it % steps_per_epoch == 0:
            lr *= 0.9; w_depth *= w_decay
            tr.set_lr(lr)
        rays, img, rgb, depth = table[it % table_batches]
        prior = torch.where(prior_mask, depth[:, 0], torch.full_like(depth[:, 0], -1.0))
        w = w_depth

        def aux(out, prior=prior, w=w):      # metrics.depth_loss_L2 (metrics.py:24-31) on the rendered depth, column 3 of the packed outputs
            valid = prior >= 0
            return ((out[:, 3][valid] - prior[valid]) ** 2).mean() * w
        tr.step(rays, img, rgb, epoch, aux_loss=aux)
        if it % 500 == 499:
            tr.check_device_status()
    tr.check_device_status()
    return f


def export_quality(f, n_batches=4, epoch=3, seed0=900):
    """Export render (module in .eval() mode under no_grad: the module's export precision -- fp16x3 for a bf16 module, fp32 for an fp32
    one) of held-out terrain rays -> per-ray altitude (datasets/satellite.py:502-533, Z_scale 50 m), its MAE against the terrain, PSNR."""
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors, get_utmalt_from_nerf_prediction
    off, sc = [0.0, 0.0, 20.0], [250.0, 250.0, Z_SCALE]
    f.eval()
    alts, gts, mse = [], [], []
    with torch.no_grad():
        for k in range(n_batches):
            rays, img, rgb, depth_gt = terrain_batch(R, seed=seed0 + k)
            g = torch.Generator(device="cuda").manual_seed(50 + k)
            noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
            res, _ = render_image(f, None, define_satrays_from_tensors(rays, img[:, None]), None, None, epoch_idx=epoch, chunk=R,
                                  render_step_size=STEP, noise=[noise])
            alts.append(get_utmalt_from_nerf_prediction(rays, res["depth"], off, sc)[2])
            gts.append(get_utmalt_from_nerf_prediction(rays, depth_gt, off, sc)[2])
            mse.append(((res["rgb"] - rgb) ** 2).mean())
    f.train()
    alt, gt = torch.cat(alts).double().flatten(), torch.cat(gts).double().flatten()
    m = torch.stack(mse).mean().item()
    import math
    return {"alt": alt, "dsm_mae_m": (alt - gt).abs().mean().item(), "psnr": -10.0 * math.log10(m)}
