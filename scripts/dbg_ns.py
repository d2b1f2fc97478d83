import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo/tests")
import torch, torch.nn.functional as F
from conftest import load_golden, T
from oracle import eonerf_oracle as orc
import test_n_samples as tn
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
for ns in (96, 192, 64):
  for tag, epoch in (("e0", 0), ("e3", 3)):
    g = load_golden(f"g10_n{ns}")
    step = float(g["step"]); sd = tn._sd(g)
    f = tn._field(sd, int(g["n_img"]), "fp32")
    rays, ts, rgbs, u_cam, u_sun = T(g["rays"]), T(g["ts"]), T(g["rgbs"]), T(g[f"{tag}.u_cam"]), T(g[f"{tag}.u_sun"])
    f.zero_grad()
    res, _ = render_image(f, None, define_satrays_from_tensors(rays.cuda(), ts.cuda()), None, None, epoch_idx=epoch, chunk=4096, render_step_size=step, noise=[(u_cam, None, u_sun)])
    pix = rgbs.cuda()
    loss = F.mse_loss(res["rgb"], pix) if epoch < 2 else ((res["rgb"] - pix) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2
    loss.backward()
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    orc.train_step(sd64, rays.double(), ts, rgbs.double(), u_cam.double(), u_sun.double(), epoch, step)
    params = dict(f.named_parameters())
    out = []
    for k, v in g.items():
        if not k.startswith(f"{tag}.grad."): continue
        name = k[len(tag) + 6:]
        p = params[name]
        got = tn.compact_grad(p.grad if p.grad is not None else torch.zeros_like(p))[2:]
        ref = T(v)[2:]
        g64 = sd64[name].grad
        r64 = tn.compact_grad(g64 if g64 is not None else torch.zeros_like(p, device="cpu"))[2:]
        ref_err, err = (ref - r64).norm().item(), (got - r64).norm().item()
        out.append((err / (2.0 * ref_err + 2e-3 * r64.norm().item() + 1e-9), name, err, ref_err, r64.norm().item()))
    out.sort(reverse=True)
    print(ns, tag, "n_samples", int(g[f"{tag}.n_samples"]), "sc", int(T(g[f"{tag}.out"])[:, 15].sum()), [ (round(a,3), n.replace("base_mlp.hidden_layers","L")) for a, n, *_ in out[:5]])
