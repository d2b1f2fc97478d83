import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo/tests")
import torch, torch.nn.functional as F
from conftest import load_golden, T
from oracle import eonerf_oracle as orc
import test_n_samples as tn
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 96
tag, epoch = "e3", 3
g = load_golden(f"g10_n{ns}")
step = float(g["step"]); sd = tn._sd(g)
rays, ts, rgbs, u_cam, u_sun = T(g["rays"]), T(g["ts"]), T(g["rgbs"]), T(g[f"{tag}.u_cam"]), T(g[f"{tag}.u_sun"])
names = ["base_mlp.hidden_layers.1.bias", "base_mlp.hidden_layers.0.weight", "sigma_layer.output_layer.weight"]
def run(sel):
    f = tn._field(sd, int(g["n_img"]), "fp32")
    f.zero_grad()
    r, t_, px, uc, us = rays[sel], ts[sel], rgbs[sel].cuda(), u_cam[sel], u_sun[sel]
    res, n = render_image(f, None, define_satrays_from_tensors(r.cuda(), t_.cuda()), None, None, epoch_idx=epoch, chunk=4096, render_step_size=step, noise=[(uc, None, us)])
    loss = ((res["rgb"] - px) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2
    loss.backward()
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    l64, out64 = orc.train_step(sd64, r.double(), t_, rgbs[sel].double(), uc.double(), us.double(), epoch, step)
    params = dict(f.named_parameters())
    rel = []
    for nm in names:
        a, b = params[nm].grad.cpu().double(), sd64[nm].grad
        rel.append(((a - b).norm() / (b.norm() + 1e-30)).item())
    o = torch.cat([res[k] for k in tn.KEYS], dim=1).cpu().double()
    return rel, (o - out64.detach()).abs().max().item(), o[:, 14:16].tolist(), float(loss), float(l64)
if len(sys.argv) > 2:      # all tensors of the named rays
    for i in [int(a) for a in sys.argv[2:]]:
        f = tn._field(sd, int(g["n_img"]), "fp32")
        f.zero_grad()
        sel = slice(i, i + 1)
        res, n = render_image(f, None, define_satrays_from_tensors(rays[sel].cuda(), ts[sel].cuda()), None, None, epoch_idx=epoch, chunk=4096, render_step_size=step, noise=[(u_cam[sel], None, u_sun[sel])])
        loss = ((res["rgb"] - rgbs[sel].cuda()) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2
        loss.backward()
        sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        orc.train_step(sd64, rays[sel].double(), ts[sel], rgbs[sel].double(), u_cam[sel].double(), u_sun[sel].double(), epoch, step)
        sd32 = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        orc.train_step(sd32, rays[sel], ts[sel], rgbs[sel], u_cam[sel], u_sun[sel], epoch, step)
        print("ray", i, "out", [round(x, 5) for x in torch.cat([res[k] for k in tn.KEYS], dim=1)[0].tolist()])
        for nm, p_ in f.named_parameters():
            b = sd64[nm].grad
            if b is None or b.norm() == 0: continue
            a = p_.grad.cpu().double(); o = sd32[nm].grad.double()
            print(f"   {nm:48s} HIP rel err {((a - b).norm() / b.norm()).item():.1e}   oracle fp32 rel err {((o - b).norm() / b.norm()).item():.1e}   norm {b.norm().item():.2e}")
    sys.exit(0)
print("all rays:", run(slice(0, 32))[:2])
for i in range(32):
    rel, fe, cnt, l, l64 = run(slice(i, i + 1))
    flag = " <<<<" if max(rel) > 1e-3 else ""
    print(i, [f"{x:.1e}" for x in rel], f"fwd err {fe:.1e}", cnt, f"loss {l:.6f} {l64:.6f}", flag)
