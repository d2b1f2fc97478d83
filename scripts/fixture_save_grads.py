import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.chdir("/root/repo/tests")
import torch
from conftest import load_golden, T
import test_n_samples as tn
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
ns = 96; g = load_golden(f"g10_n{ns}"); step = float(g["step"]); sd = tn._sd(g)
rays, ts, rgbs, u_cam, u_sun = T(g["rays"]), T(g["ts"]), T(g["rgbs"]), T(g["e3.u_cam"]), T(g["e3.u_sun"])
out = {}
for i in (26, 15, 23, 6):
    sel = slice(i, i + 1)
    f = tn._field(sd, int(g["n_img"]), "fp32")
    f.zero_grad()
    res, n = render_image(f, None, define_satrays_from_tensors(rays[sel].cuda(), ts[sel].cuda()), None, None, epoch_idx=3, chunk=4096, render_step_size=step, noise=[(u_cam[sel], None, u_sun[sel])])
    loss = ((res["rgb"] - rgbs[sel].cuda()) ** 2 / (2 * res["beta"] ** 2)).mean() + (3 + torch.log(res["beta"]).mean()) / 2
    loss.backward()
    out[i] = {k: p.grad.cpu() for k, p in f.named_parameters() if p.grad is not None}
torch.save(out, "/root/repo/gpurun_out/r6r/hip_grads.pt")
print("saved")
