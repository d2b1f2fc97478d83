import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from test_hip_backward import make_field, hip_step, oracle_step
from oracle import eonerf_oracle as orc
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_img, Rn = 6, 192
sd = orc.random_state_dict(n_img, seed=51, bias_scale=0.05, radiometric_jitter=0.05)
sd["sigma_layer.output_layer.bias"] += 1.0
rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(Rn, n_img, seed=52)
f = make_field(sd, n_img, prec)
loss, _ = hip_step(f, rays, ts, rgbs, (u_cam, None, u_sun), epoch)
ref_loss, ref = oracle_step(sd, rays, ts, rgbs, u_cam, u_sun, epoch)
print("loss", loss.item(), ref_loss.item())
for name, p in f.named_parameters():
    rg = ref[name] if ref[name] is not None else torch.zeros_like(sd[name])
    got = p.grad.cpu() if p.grad is not None else torch.zeros_like(rg)
    scale = rg.abs().max().item()
    err = (got - rg).abs()
    line = f"{name:45s} scale {scale:.3e} maxerr {err.max().item():.3e} rel {err.max().item()/(scale+1e-30):.2e}"
    if got.dim() == 2 and err.max().item() > 1e-3 * scale + 1e-7:
        bad_cols = (err.max(dim=0).values > 1e-3 * scale).nonzero().flatten().tolist()
        bad_rows = (err.max(dim=1).values > 1e-3 * scale).nonzero().flatten().tolist()
        line += f"  bad cols {len(bad_cols)}/{got.shape[1]} {bad_cols[:12]} bad rows {len(bad_rows)}/{got.shape[0]} {bad_rows[:8]}"
    print(line)
