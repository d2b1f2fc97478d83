"""GPU parity of the HIP backward path (autograd of render_image through the C-ABI) against the reference's own
autograd (golden G8 gradients) and torch autograd on the oracle.

Tolerances
  fp32 mode: every parameter gradient within 2e-3 relative L2 error of the oracle's and within 1e-2 * max|g| per
      element.  Both sides are fp32 FMA chains; the residual is NOT rounding noise of the sums but ReLU-derivative
      flips: a pre-activation within ~1e-6 of zero takes the other branch of relu'() under a different (equally valid)
      fp32 summation order, which changes one (unit, sample) contribution by O(1) -- visible as isolated rows of a dW.
      Tensors untouched by such a flip agree to ~3e-6 relative.
      With the shadow pass on, d sigma/d position runs through the encoder derivative 2^k cos(2^k x) (k <= 9) and the
      reference's OWN fp32 autograd is only good to a few % against an fp64 evaluation of the same graph (measured:
      2-4 % on the G8 network).  There the criterion is "as exact as the reference's arithmetic":
          |hip - fp64| <= 1.5 |fp32 reference - fp64| + 2e-3 |fp64|      (L2 norms per tensor; measured worst ratio in DESIGN.md 5).
  bf16 mode: gradients are bf16-rounded at every layer boundary; checked per tensor by relative L2 error < 6e-2 and
      cosine similarity > 0.998 against the fp32 oracle.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, T
from oracle import eonerf_oracle as orc

pytestmark = pytest.mark.gpu
STEP = 2.0 / 128
GRAD_STRIDE = 61


def compact_grad(g):
    flat = g.detach().reshape(-1).double().cpu()
    head = torch.stack([flat.sum(), flat.abs().sum()])
    body = flat if flat.numel() <= 1024 else flat[::GRAD_STRIDE]
    return torch.cat([head, body])


def make_field(sd, n_img, precision):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    f = EONerfMLP(n_img, radiometric_normalization=True, precision=precision)
    f.load_state_dict(sd, strict=True)
    return f.cuda()


def hip_step(f, rays, ts, rgbs, noise, epoch):
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    f.zero_grad()
    res, n = render_image(f, None, define_satrays_from_tensors(rays.cuda(), ts.cuda()), None, None, epoch_idx=epoch, chunk=4096,
                          render_step_size=STEP, noise=[noise])
    pix = rgbs.cuda()
    if epoch < 2:
        loss = F.mse_loss(res["rgb"], pix)
    else:   # metrics.uncertainty_aware_loss (metrics.py:17-22), plain torch ops on [R,3]
        loss = ((res["rgb"] - pix) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2
    loss.backward()
    return loss.detach().cpu(), res


def oracle_step(sd, rays, ts, rgbs, u_cam, u_sun, epoch):
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    loss, out = orc.train_step(sdg, rays, ts, rgbs, u_cam, u_sun, epoch, STEP)
    return loss, {k: v.grad for k, v in sdg.items() if v.is_floating_point()}


def oracle_step64(sd, rays, ts, rgbs, u_cam, u_sun, epoch):
    sdg = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    loss, _ = orc.train_step(sdg, rays.double(), ts, rgbs.double(), u_cam.double(), u_sun.double(), epoch, STEP)
    return {k: v.grad for k, v in sdg.items() if v.is_floating_point()}


# "as exact as the reference's own fp32 autograd": |hip - fp64| <= EXACT_FACTOR |ref32 - fp64| + 2e-3 |fp64| per tensor.
# Measured on MI355X (round 2, printed by test_zz_report_exactness_ratios): the worst ratio err / (ref_err + 2e-3 |fp64|) over all
# tensors and cases is recorded in DESIGN.md 5; the factor was 3 in round 1.
EXACT_FACTOR = 1.5
RATIOS = []


def check_as_exact_as_reference(f, ref32, ref64, tag):
    for name, p in f.named_parameters():
        r64 = ref64[name] if ref64[name] is not None else torch.zeros_like(p, dtype=torch.float64, device="cpu")
        r32 = (ref32[name] if ref32[name] is not None else torch.zeros_like(r64)).double()
        got = (p.grad.cpu() if p.grad is not None else torch.zeros_like(r64)).double()
        ref_err = (r32 - r64).norm().item()
        err = (got - r64).norm().item()
        RATIOS.append((err / (ref_err + 2e-3 * r64.norm().item() + 1e-9), tag, name))
        assert err <= EXACT_FACTOR * ref_err + 2e-3 * r64.norm().item() + 1e-9, (tag, name, err, ref_err, r64.norm().item())


@pytest.mark.parametrize("tag,epoch", [("e0", 0), ("e3", 3)])
def test_backward_fp32_matches_reference_autograd_g8(tag, epoch):
    g = load_golden("g8_render")
    sd = orc.closed_form_state_dict(int(g["n_img"]))
    sd["sigma_layer.output_layer.bias"] = sd["sigma_layer.output_layer.bias"] + float(g["sigma_bias_shift"])
    f = make_field(sd, int(g["n_img"]), "fp32")
    rays, ts, rgbs, u_cam, u_sun = T(g["rays"]), T(g["ts"]), T(g["rgbs"]), T(g[f"{tag}.u_cam"]), T(g[f"{tag}.u_sun"])
    loss, _ = hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), epoch)
    assert abs(loss.item() - float(g[f"{tag}.loss"])) < 1e-5
    ref64 = oracle_step64(sd, rays, ts, rgbs, u_cam, u_sun, epoch)
    params = dict(f.named_parameters())
    for k, v in g.items():       # golden = the reference's own fp32 autograd (strided subsample of every gradient)
        if not k.startswith(f"{tag}.grad."):
            continue
        name = k[len(tag) + 6:]
        p = params[name]
        got = compact_grad(p.grad if p.grad is not None else torch.zeros_like(p))[2:]
        ref = T(v)[2:]
        r64 = compact_grad(ref64[name] if ref64[name] is not None else torch.zeros_like(p, device="cpu"))[2:]
        ref_err = (ref - r64).norm().item()
        err = (got - r64).norm().item()
        RATIOS.append((err / (ref_err + 2e-3 * r64.norm().item() + 1e-9), tag, name))
        assert err <= EXACT_FACTOR * ref_err + 2e-3 * r64.norm().item() + 1e-9, (k, err, ref_err)
        if epoch < 2:    # no encoder-derivative path: direct agreement with the reference's fp32 numbers
            assert (got - ref).norm().item() <= 2e-3 * ref.norm().item() + 1e-7, k


def test_backward_fp32_random_batch_all_parameters():
    n_img, R = 6, 192
    sd = orc.random_state_dict(n_img, seed=51, bias_scale=0.05, radiometric_jitter=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, n_img, seed=52)
    for epoch in (0, 3):
        f = make_field(sd, n_img, "fp32")
        loss, _ = hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), epoch)
        ref_loss, ref = oracle_step(sd, rays, ts, rgbs, u_cam, u_sun, epoch)
        assert abs(loss.item() - ref_loss.item()) < 1e-5
        check_as_exact_as_reference(f, ref, oracle_step64(sd, rays, ts, rgbs, u_cam, u_sun, epoch), f"epoch{epoch}")
        for name, p in f.named_parameters():
            rg = ref[name] if ref[name] is not None else torch.zeros_like(sd[name])
            got = p.grad.cpu() if p.grad is not None else torch.zeros_like(rg)
            assert (got - rg).abs().max().item() <= 1e-2 * rg.abs().max().item() + 1e-7, (epoch, name)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config4_twenty_dates_learned_radiometric_correction_step(precision):
    """BASELINE.json configs[4] on one GPU: 20 acquisition dates, every date with its own learned radiometric affine (A, b rows away
    from the identity: opt.py:98-99, sat_rendering.py:288-306) and transient embedding, full EO-NeRF step (shadow pass + uncertainty
    loss), every image index present in the batch.  fp32: outputs 1e-4, every parameter gradient -- the 20 x 9 radiometric table and
    the 20 x 4 embedding included -- 2e-3 relative L2 of torch autograd on the oracle ("RPC bundle-adjust parameters" do not exist in
    the shipped reference).  bf16: against the oracle's bf16 arithmetic model, 3e-2 / cosine 0.999."""
    n_img, R = 20, 320
    sd = orc.random_state_dict(n_img, seed=131, bias_scale=0.05, radiometric_jitter=0.1)
    sd["sigma_layer.output_layer.bias"] += 1.0
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, n_img, seed=132)
    ts[:n_img, 0] = torch.arange(n_img)                                   # every date is in the batch (a ray carries its own sun direction)
    f = make_field(sd, n_img, precision)
    loss, res = hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), 3)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ref_loss, ref_out = orc.train_step(sdg, rays, ts, rgbs, u_cam, u_sun, 3, STEP, emulate_bf16=(precision == "bf16"))
    assert abs(loss.item() - ref_loss.item()) < (1e-5 if precision == "fp32" else 2e-3)
    if precision == "fp32":
        assert (res["rgb"].cpu() - ref_out[:, 0:3]).abs().max().item() < 1e-4 and (res["shadowless_rgb"].cpu() - ref_out[:, 18:21]).abs().max().item() < 1e-4
    for name, p in f.named_parameters():
        rg = sdg[name].grad
        assert rg is not None and rg.norm() > 0, name                     # everything is in the graph in this configuration
        got = p.grad.cpu()
        rel = ((got - rg).norm() / rg.norm()).item()
        cos = ((got * rg).sum() / (got.norm() * rg.norm())).item()
        assert (rel < 2e-3) if precision == "fp32" else (rel < 3e-2 and cos > 0.999), (name, rel, cos)
    rad = dict(f.named_parameters())["radiometricT_enc.weight"].grad
    assert (rad[:, :6].abs().sum(dim=1) > 0).all() and rad[:, 6:].abs().max().item() == 0.0      # all 20 rows of A | b, the 3 unused columns untouched


def test_backward_bf16_close_to_fp32_autograd():
    n_img, R = 5, 256
    sd = orc.random_state_dict(n_img, seed=61, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, n_img, seed=62)
    f = make_field(sd, n_img, "bf16")
    hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), 3)
    _, ref = oracle_step(sd, rays, ts, rgbs, u_cam, u_sun, 3)
    for name, p in f.named_parameters():
        rg = ref[name]
        if rg is None or rg.abs().max() == 0:
            continue
        got = p.grad.cpu().flatten().double()
        r = rg.flatten().double()
        rel = (got - r).norm() / r.norm()
        cos = torch.dot(got, r) / (got.norm() * r.norm())
        assert rel < 6e-2 and cos > 0.998, (name, rel.item(), cos.item())


def test_adam_step_matches_torch():
    import ctypes as C
    from eonerf_code_amd import _lib
    from eonerf_code_amd.radiance_fields.eonerf import _ptr, _stream
    sd = orc.random_state_dict(3, seed=71)
    f = make_field(sd, 3, "fp32")
    flat = f.flat_params()
    g = torch.Generator(device="cpu").manual_seed(1)
    ref_p = flat.detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=5e-4)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    for step in range(1, 4):
        grad = torch.randn(flat.numel(), generator=g) * 0.01
        ref_p.grad = grad.clone()
        opt.step()
        _lib.check(_lib.lib().eonerf_adam_step(f._context(), _ptr(flat), _ptr(grad.cuda()), _ptr(m), _ptr(v), step,
                                               5e-4, 0.9, 0.999, 1e-8, 1.0, None, _stream()))
    assert (flat.cpu() - ref_p.detach()).abs().max().item() < 1e-6


def test_training_reduces_loss_bf16():
    n_img, R = 4, 512
    sd = orc.random_state_dict(n_img, seed=81)
    rays, ts, rgbs, _, _ = orc.synthetic_batch(R, n_img, seed=82)
    rgbs = 0.5 + 0.2 * torch.sin(rays[:, :3] * 3.0)     # a learnable target
    f = make_field(sd, n_img, "bf16")
    opt = torch.optim.Adam(f.parameters(), lr=5e-4)
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    satrays = define_satrays_from_tensors(rays.cuda(), ts.cuda())
    pix = rgbs.cuda()
    losses = []
    for it in range(30):
        res, _ = render_image(f, None, satrays, None, None, epoch_idx=0, chunk=4096, render_step_size=STEP)
        loss = F.mse_loss(res["rgb"], pix)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], losses


def test_fused_trainer_matches_autograd_path_and_oracle():
    # FusedTrainer.step (direct C calls, fused loss kernel, EONERF_F_RGB_LOSS graph pruning for epoch < 2) computes the same
    # loss and the same gradients as the autograd path.  Gradients are compared, not post-Adam weights: Adam's first step is
    # lr*g/(|g|+eps), which turns atomic-order noise on |g| ~ eps into O(lr) differences.
    from eonerf_code_amd.trainer import FusedTrainer
    n_img, R = 4, 128
    sd = orc.random_state_dict(n_img, seed=91, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, n_img, seed=92)
    for epoch in (0, 3):
        f1, f2 = make_field(sd, n_img, "fp32"), make_field(sd, n_img, "fp32")
        loss1, _ = hip_step(f1, rays, ts, rgbs, (u_cam, None, u_sun), epoch)
        tr = FusedTrainer(f2, lr=5e-4, max_rays=R)
        before = f2.flat_params().clone()
        loss2 = tr.step(rays.cuda(), ts.reshape(-1).cuda(), rgbs.cuda(), epoch, noise=(u_cam.cuda(), None, u_sun.cuda()))
        assert abs(loss1.item() - float(loss2)) < 1e-5
        for (name, p1), g2 in zip(f1.named_parameters(), f2.grad_views(tr.d_flat)):
            g1 = p1.grad if p1.grad is not None else torch.zeros_like(p1)
            assert (g1 - g2).norm().item() <= 1e-4 * g1.norm().item() + 1e-9, (epoch, name)
        moved = (f2.flat_params() - before).abs().max().item()
        assert 0 < moved <= 5e-4 * 1.001                              # one Adam step moves a weight by at most lr


def test_zz_report_exactness_ratios():
    """Runs last in this file: prints the worst measured ratio err / (ref_err + 2e-3 |fp64|) of the fp32 gradient checks above."""
    if not RATIOS:
        pytest.skip("the fp32 gradient tests did not run in this session")
    worst = sorted(RATIOS, reverse=True)[:5]
    print("worst exactness ratios (err / (ref_err + 2e-3 |fp64|)):", [(round(r, 3), t, n) for r, t, n in worst])
    assert worst[0][0] <= EXACT_FACTOR


def test_iarpa_like_configuration_twenty_images_radiometric_fp32():
    """BASELINE.json configs[4] (IARPA, 20 dates, learned radiometric correction) at a size the oracle evaluates in seconds: n_img = 20,
    radiometric table away from its identity init, shadow pass on -- outputs and every gradient incl. the 20 x 9 radiometric table and the
    20 x 4 transient embedding against torch autograd on the oracle.  ("RPC bundle-adjust params" do not exist in the shipped reference,
    SURVEY.md 0.)"""
    n_img, R = 20, 512
    sd = orc.random_state_dict(n_img, seed=77, bias_scale=0.05, radiometric_jitter=0.1)
    sd["sigma_layer.output_layer.bias"] += 1.0
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, n_img, seed=78)
    f = make_field(sd, n_img, "fp32")
    loss, res = hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), 3)
    ref_loss, ref = oracle_step(sd, rays, ts, rgbs, u_cam, u_sun, 3)
    with torch.no_grad():
        out, _ = orc.render_rays(orc.Field(sd), orc.define_satrays_from_tensors(rays, ts), u_cam, u_sun, 3, STEP)
    assert (res["rgb"].detach().cpu() - out[:, 0:3]).abs().max().item() < 1e-4
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    check_as_exact_as_reference(f, ref, oracle_step64(sd, rays, ts, rgbs, u_cam, u_sun, 3), "iarpa20")
    params = dict(f.named_parameters())
    for name in ("radiometricT_enc.weight", "transient_encoder.weight"):
        g, r = params[name].grad.cpu(), ref[name]
        assert tuple(g.shape) == (n_img, 9 if "radio" in name else 4)
        assert (g - r).norm().item() <= 2e-3 * r.norm().item() + 1e-9, name
