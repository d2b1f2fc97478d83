"""Ray 26 of the n_samples = 96 fixture: the trunk's INPUT gradient at the shadow ray's samples, HIP fp32 (differentiable query_density)
against the oracle in fp64 and fp32, cotangent = 1 for every sample and a random one."""
import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo/tests")
import torch
from conftest import load_golden, T
from oracle import eonerf_oracle as orc
import test_n_samples as tn
ns = 96; g = load_golden(f"g10_n{ns}"); step = float(g["step"]); sd = tn._sd(g)
rays, ts, u_cam, u_sun = T(g["rays"]), T(g["ts"]), T(g["e3.u_cam"]), T(g["e3.u_sun"])
for i in (26, 15, 6):
    sel = slice(i, i + 1)
    r = orc.define_satrays_from_tensors(rays[sel], ts[sel])
    field = orc.Field(sd)
    with torch.no_grad():
        ri, a, b = orc.satnerf_sampling(r.origins, r.viewdirs, u_cam[sel], step, near=r.t_near)
        albedo, depth, *_ = orc.rendering(field, r, a, b.clone(), ri)
        so = r.origins + depth * r.viewdirs
        sri, sa, sb = orc.satnerf_sampling(so, -r.sundirs, u_sun[sel], step)
        pos = so[sri] + (-r.sundirs)[sri] * ((sa + sb)[:, None] / 2)
    gen = torch.Generator().manual_seed(3)
    for tag, cot in (("ones", torch.ones(pos.shape[0], 1)), ("random", torch.randn(pos.shape[0], 1, generator=gen))):
        res = {}
        for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
            x = pos.to(dt).clone().requires_grad_(True)
            s = orc.Field(sdd).query_density(x)
            (s * cot.to(dt)).sum().backward()
            res[name] = x.grad.double()
        f = tn._field(sd, int(g["n_img"]), "fp32")
        x = pos.cuda().clone().requires_grad_(True)
        s = f.query_density(x)
        (s * cot.cuda()).sum().backward()
        hip = x.grad.cpu().double()
        n = res["f64"].norm()
        print(f"ray {i} {tag}: {pos.shape[0]} sun samples, |dx| {n:.3e}; HIP rel err {(hip - res['f64']).norm() / n:.2e}, oracle fp32 rel err {(res['f32'] - res['f64']).norm() / n:.2e}; "
              f"worst sample HIP {((hip - res['f64']).norm(dim=1) / res['f64'].norm(dim=1)).max():.2e} oracle32 {((res['f32'] - res['f64']).norm(dim=1) / res['f64'].norm(dim=1)).max():.2e}")
