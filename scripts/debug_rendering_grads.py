"""per-tensor gradient error of EONerfMLP.rendering()/render_depth() under autograd against torch autograd on the oracle (fp32)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import eonerf_oracle as orc
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
n_img, R, STEP = 4, 96, 2.0 / 128
sd = orc.random_state_dict(n_img, seed=111, bias_scale=0.05)
sd["sigma_layer.output_layer.bias"] += 1.0
rays, ts, _, u_cam, _ = orc.synthetic_batch(R, n_img, seed=112)
orays = orc.define_satrays_from_tensors(rays, ts)
ri, a, b = orc.satnerf_sampling(orays.origins, orays.viewdirs, u_cam, STEP, near=orays.t_near)
g = torch.Generator().manual_seed(5)
cot = [torch.randn(R, c, generator=g) for c in (3, 1, 1, 1, 3)]
res = {}
for dt in (torch.float32, torch.float64):
    sdg = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    r_ = orc.define_satrays_from_tensors(rays.to(dt), ts)
    ref = orc.rendering(orc.Field(sdg), r_, a.to(dt), b.to(dt), ri)
    sum((r * c.to(dt)).sum() for r, c in zip(ref[:5], cot)).backward()
    res[dt] = {k: v.grad for k, v in sdg.items() if v.is_floating_point()}
f = EONerfMLP(n_img, radiometric_normalization=True, precision="fp32")
f.load_state_dict(sd, strict=True); f = f.cuda()
hrays = define_satrays_from_tensors(rays.cuda(), ts.cuda())
got = f.rendering(hrays, a.cuda(), b.clone().cuda(), ri.cuda())
sum((h * c.cuda()).sum() for h, c in zip(got[:5], cot)).backward()
for name, p in f.named_parameters():
    r64 = res[torch.float64][name]
    if r64 is None or r64.norm() == 0: continue
    r32 = res[torch.float32][name].double()
    e_hip = ((p.grad.cpu().double() - r64).norm() / r64.norm()).item()
    e_ref = ((r32 - r64).norm() / r64.norm()).item()
    print(f"{name:45s} hip-vs-fp64 {e_hip:.2e}   oracle32-vs-fp64 {e_ref:.2e}   |g| {r64.norm().item():.3e}")
