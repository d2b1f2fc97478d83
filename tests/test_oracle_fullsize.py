"""GPU: oracle parity AT THE BENCHMARKED SIZE -- bench.py's own synthetic batch (4096 rays x 128 samples, 19 images, Xavier weights),
both benchmarked configurations (BASELINE.json configs[1]: epoch_idx < 2, configs[2]: shadow pass on).

  fp32-HIP vs the oracle (sat_rendering.py:252-312, radiance_fields/eonerf.py:196-248 restated): all 21 output columns <= 1e-4,
      sample bookkeeping bit exact, gradients by the "as exact as the reference's own fp32 autograd" criterion of
      tests/test_hip_backward.py (|hip - fp64| <= 1.5 |oracle32 - fp64| + 2e-3 |fp64| per tensor);
  bf16-HIP forward vs the bf16-emulating oracle (same rounding points) at the tolerances of tests/test_hip_forward.py;
  backward through a MULTI-CHUNK render_image call (chunk=1024 < batch, the reference's training call: sat_rendering.py:252 with
      opt.py:60) -- four outstanding workspaces, pipelined path -- against the single-chunk gradient;
  EONerfMLP.query_opacity on the 128^3 occupancy grid of train_eonerf.py:112-119 (opt.py:86): one 2,097,152-point call against a
      chunked call (bit identical) and the oracle on a subsample.

The oracle runs in ray chunks (its loss is a sum over rays, so the gradient is the sum of the chunks' gradients): a 4096-ray step
in fp64 would otherwise hold ~25 GB of autograd state on the host.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import eonerf_oracle as orc

pytestmark = pytest.mark.gpu
STEP = 2.0 / 128
N_IMG, R = 19, 4096
EXACT_FACTOR = 1.5


@pytest.fixture(scope="module")
def batch():
    sd = orc.random_state_dict(N_IMG, seed=42)                       # bench.py's cpu_baseline weights
    return (sd,) + tuple(orc.synthetic_batch(R, N_IMG, seed=1234))   # ... and batch


def make_field(sd, precision):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision=precision)
    f.load_state_dict(sd, strict=True)
    return f.cuda()


def loss_sum_form(out, pix, epoch, n_total):
    """train_eonerf.py:139-143 written as a sum over rays divided by the size of the WHOLE batch (chunks add up)."""
    if epoch < 2:
        return ((out[:, 0:3] - pix) ** 2).sum() / (3 * n_total)
    beta = out[:, 12:13]
    return ((out[:, 0:3] - pix) ** 2 / (2 * beta ** 2)).sum() / (3 * n_total) + torch.log(beta).sum() / (2 * n_total)


def oracle_chunked(sd, rays, ts, rgbs, u_cam, u_sun, epoch, dtype, emulate_bf16=False, chunk=512, grads=True):
    """Outputs [R,21] and (optionally) parameter gradients of the oracle, ray chunk by ray chunk."""
    cast = (lambda t: t.to(dtype)) if dtype != torch.float32 else (lambda t: t)
    sdg = {k: (cast(v).clone().requires_grad_(grads) if v.is_floating_point() else v) for k, v in sd.items()}
    outs, n_tot, loss_tot = [], 0, 0.0
    for i in range(0, rays.shape[0], chunk):
        sl = slice(i, i + chunk)
        field = orc.Field(sdg, emulate_bf16)
        with torch.set_grad_enabled(grads):
            out, n = orc.render_rays(field, orc.define_satrays_from_tensors(cast(rays[sl]), ts[sl]), cast(u_cam[sl]), cast(u_sun[sl]), epoch, STEP)
            if grads:
                loss = loss_sum_form(out, cast(rgbs[sl]), epoch, rays.shape[0])
                loss.backward()
                loss_tot += float(loss.detach())
        outs.append(out.detach())
        n_tot += n
    g = {k: v.grad for k, v in sdg.items() if v.is_floating_point()} if grads else None
    return torch.cat(outs), n_tot, loss_tot + (1.5 if epoch >= 2 else 0.0), g


def hip_render(f, rays, ts, noise_chunks, epoch, chunk):
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    return render_image(f, None, define_satrays_from_tensors(rays.cuda(), ts.cuda()), None, None, epoch_idx=epoch, chunk=chunk,
                        render_step_size=STEP, noise=noise_chunks)


def hip_loss(res, pix, epoch):
    if epoch < 2:
        return F.mse_loss(res["rgb"], pix)
    return ((res["rgb"] - pix) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2


def noise_for(u_cam, u_sun, chunk):
    return [(u_cam[i:i + chunk], None, u_sun[i:i + chunk]) for i in range(0, u_cam.shape[0], chunk)]


@pytest.mark.parametrize("epoch", [0, 3])
def test_fp32_outputs_and_gradients_match_the_oracle_at_the_bench_size(batch, epoch):
    sd, rays, ts, rgbs, u_cam, u_sun = batch
    ref, n_ref, loss32, g32 = oracle_chunked(sd, rays, ts, rgbs, u_cam, u_sun, epoch, torch.float32)
    _, _, _, g64 = oracle_chunked(sd, rays, ts, rgbs, u_cam, u_sun, epoch, torch.float64)
    f = make_field(sd, "fp32")
    f.zero_grad()
    res, n = hip_render(f, rays, ts, noise_for(u_cam, u_sun, R), epoch, R)
    loss = hip_loss(res, rgbs.cuda(), epoch)
    loss.backward()
    # ---- forward: 1e-4 on every column (BASELINE.json: sigma/rgb within 1e-4 fp32), bookkeeping bit exact ----
    assert n == n_ref
    for name, (a, b) in orc.RESULT_SLICES.items():
        got, want = res[name].cpu(), ref[:, a:b]
        if name in ("pts_per_ray", "sc_pts_per_ray", "entropy", "opacity_after_surface"):
            assert torch.equal(got, want), name
        else:
            assert (got - want).abs().max().item() < 1e-4, (name, (got - want).abs().max().item())
    assert abs(float(loss) - loss32) < 1e-5 * max(1.0, abs(loss32))
    # ---- backward: as exact as the reference's own fp32 autograd, per tensor ----
    worst = 0.0
    for name, p in f.named_parameters():
        r64 = g64[name] if g64[name] is not None else torch.zeros_like(p, dtype=torch.float64, device="cpu")
        r32 = (g32[name] if g32[name] is not None else torch.zeros_like(r64)).double()
        got = (p.grad.cpu() if p.grad is not None else torch.zeros_like(r64)).double()
        ref_err, err = (r32 - r64).norm().item(), (got - r64).norm().item()
        bound = EXACT_FACTOR * ref_err + 2e-3 * r64.norm().item() + 1e-9
        worst = max(worst, err / bound)
        assert err <= bound, (epoch, name, err, ref_err, r64.norm().item())
    print(f"[fullsize fp32 epoch {epoch}] worst err / bound = {worst:.3f}")


_BF16_ORACLE = {}


def oracle_bf16(batch, epoch):
    """ONE run of the oracle's bf16 arithmetic model per epoch (outputs AND gradients; ~40 s of host time), shared by the forward and
    the backward test."""
    if epoch not in _BF16_ORACLE:
        sd, rays, ts, rgbs, u_cam, u_sun = batch
        _BF16_ORACLE[epoch] = oracle_chunked(sd, rays, ts, rgbs, u_cam, u_sun, epoch, torch.float32, emulate_bf16=True, grads=True)
    return _BF16_ORACLE[epoch]


# per-tensor relative L2 bound of the bf16 HIP gradient against the bf16 arithmetic model's autograd at the bench size: measured values
# x 2 (printed by the test), by tensor family.  What is left between the two is the order of the fp32 sums, where a
# gradient is rounded to bf16 (the model rounds dX behind every product, the kernels round dY in front of the next one -- the same
# tensor, except at the heads' outputs) and __sinf / __cosf.
BF16_GRAD_BOUNDS = {"base_mlp.hidden_layers.0.weight": 3e-2,       # measured 1.3e-2 (epoch 0) / 4.9e-3 (epoch 3)
                    "default": 1.3e-2}                               # measured <= 6.2e-3 (hidden_layers.1.weight, epoch 0); round 3's gate: 6e-2


@pytest.mark.parametrize("epoch", [0, 3])
def test_bf16_backward_matches_the_bf16_arithmetic_model_at_the_bench_size(batch, epoch):
    sd, rays, ts, rgbs, u_cam, u_sun = batch
    _, _, loss_ref, g_ref = oracle_bf16(batch, epoch)
    f = make_field(sd, "bf16")
    f.zero_grad()
    res, n = hip_render(f, rays, ts, noise_for(u_cam, u_sun, R), epoch, R)
    loss = hip_loss(res, rgbs.cuda(), epoch)
    loss.backward()
    assert abs(float(loss) - loss_ref) < 2e-3 * max(1.0, abs(loss_ref)), (float(loss), loss_ref)
    rows = []
    for name, p in f.named_parameters():
        r = g_ref[name]
        if r is None or r.norm().item() == 0.0:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        got = p.grad.cpu()
        rel = ((got - r).norm() / r.norm()).item()
        cos = ((got * r).sum() / (got.norm() * r.norm())).item()
        rows.append((rel, cos, name))
        bound = next((v for k, v in BF16_GRAD_BOUNDS.items() if k != "default" and name.startswith(k)), BF16_GRAD_BOUNDS["default"])
        assert rel < bound and cos > 0.9995, (epoch, name, rel, cos)
    rows.sort(reverse=True)
    print(f"[fullsize bf16 backward epoch {epoch}] worst per-tensor rel L2 / cosine: " + ", ".join(f"{n} {r:.2e}/{c:.5f}" for r, c, n in rows[:6]))


# Bounds = ~5 x the values measured on MI355X (round 5, printed by the test: worst column max 2.2e-4, p99.9 1.3e-4, mean 1.4e-5); the
# round-4 gate was max 3e-2 / mean 1e-3, wide enough at the tail to hide a single-ray defect
MAX_BOUND, P999_BOUND, MEAN_BOUND = 1.5e-3, 6e-4, 7e-5


@pytest.mark.parametrize("epoch", [0, 3])
def test_bf16_forward_matches_the_bf16_emulating_oracle_at_the_bench_size(batch, epoch):
    sd, rays, ts, rgbs, u_cam, u_sun = batch
    ref, n_ref, _, _ = oracle_bf16(batch, epoch)
    f = make_field(sd, "bf16")
    with torch.no_grad():
        res, n = hip_render(f, rays, ts, noise_for(u_cam, u_sun, R), epoch, R)
    assert n == n_ref
    assert torch.equal(res["pts_per_ray"].cpu(), ref[:, 14:15])
    stats = []
    for name in ("rgb", "albedo_rgb", "shadowless_rgb", "transient_s", "geo_shadows", "depth", "beta", "ambient_rgb"):
        a, b = orc.RESULT_SLICES[name]
        d = (res[name].cpu() - ref[:, a:b]).abs()
        # same rounding points on both sides; what is left is the summation order inside a bf16-input dot product and __sinf.  The MEAN
        # and the 99.9th percentile are the bounds that would catch a defect; the max is an outlier bound (one ray in 4096 whose surface
        # sample flips a bf16 rounding of sigma): a single-ray defect shows in p99.9 only if it hits > 4 rays, in the max always
        p999 = d.flatten().kthvalue(max(1, int(0.999 * d.numel()))).values.item()
        stats.append(f"{name} max {d.max().item():.1e} p99.9 {p999:.1e} mean {d.mean().item():.1e}")
        assert d.max().item() < MAX_BOUND and p999 < P999_BOUND and d.mean().item() < MEAN_BOUND, (name, d.max().item(), p999, d.mean().item())
    print(f"[fullsize bf16 forward epoch {epoch}] " + "; ".join(stats))
    if epoch >= 2:      # rays whose shadow ray keeps/loses a sample at the cube face differ by a whole sample: rare
        assert (res["sc_pts_per_ray"].cpu() != ref[:, 15:16]).float().mean().item() < 0.02


@pytest.mark.parametrize("precision,epoch", [("bf16", 0), ("bf16", 3), ("fp32", 3)])
def test_backward_through_a_multi_chunk_render_image_call(batch, precision, epoch):
    """The reference's training call: chunk (opt.py:60, default 1024) smaller than the batch -- four _RenderChunk workspaces alive at
    once, each with its own rings / sync block, backward in reverse order.  Per-ray arithmetic is identical to the single-chunk call
    (outputs bit identical); gradients differ by the summation order only."""
    sd, rays, ts, rgbs, u_cam, u_sun = batch
    pix = rgbs.cuda()
    grads, outs = [], []
    for chunk in (R, 1024):
        f = make_field(sd, precision)
        f.zero_grad()
        res, n = hip_render(f, rays, ts, noise_for(u_cam, u_sun, chunk), epoch, chunk)
        hip_loss(res, pix, epoch).backward()
        grads.append({name: p.grad.detach().clone() for name, p in f.named_parameters() if p.grad is not None})
        outs.append((n, torch.cat([res[k] for k, _, _ in __import__("eonerf_code_amd.sat_rendering", fromlist=["x"]).RESULT_SLICES], dim=1)))
        from eonerf_code_amd import _lib
        from eonerf_code_amd.radiance_fields.eonerf import _stream
        _lib.check(_lib.lib().eonerf_device_status(f._ctx, _stream()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    for name, a in grads[0].items():
        b = grads[1][name]
        assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10, (name, (a - b).norm().item(), a.norm().item())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_query_opacity_on_the_128_cubed_occupancy_grid(batch, precision):
    """train_eonerf.py:112-119: occupancy_grid.update_every_n_steps calls query_opacity(x, render_step_size) on the cells of a 128^3
    grid (opt.py:86) -- 2,097,152 points in one call."""
    sd = batch[0]
    f = make_field(sd, precision)
    n = 128
    g = (torch.arange(n, dtype=torch.float32) + 0.5) / n * 2 - 1                  # cell centres of the [-1,1]^3 aabb
    x = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3).cuda()
    with torch.no_grad():
        occ = f.query_opacity(x, STEP)
        assert occ.shape == (n ** 3, 1) and torch.isfinite(occ).all()
        parts = torch.cat([f.query_opacity(x[i:i + 262144], STEP) for i in range(0, n ** 3, 262144)])
        assert torch.equal(occ, parts)                                            # per-point arithmetic: chunking changes nothing
        idx = torch.randperm(n ** 3, generator=torch.Generator().manual_seed(5))[:4096]
        ref = orc.Field(sd, emulate_bf16=(precision == "bf16")).query_opacity(x[idx.cuda()].cpu(), STEP)
    err = (occ[idx.cuda()].cpu() - ref).abs()
    if precision == "fp32":
        assert err.max().item() < 1e-4 * STEP + 1e-7, err.max().item()           # sigma within 1e-4, times the step
    else:
        assert (err / (ref.abs() + 1e-3)).max().item() < 2e-2, (err / (ref.abs() + 1e-3)).max().item()
