"""Diagnostic: average launch time of the chain kernels of the library named by EONERF_LIB (inference + training)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.trainer import FusedTrainer
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
dev = torch.device("cuda")
torch.manual_seed(42)
f = EONerfMLP(19, radiometric_normalization=True, precision="bf16").to(dev)
rays, img, rgbs = (t.to(dev) for t in synthetic_batch(4096, 19))
tr = FusedTrainer(f)
sr = define_satrays_from_tensors(rays, img[:, None])
def infer(n):
    with torch.no_grad():
        for _ in range(n): render_image(f, None, sr, None, None, epoch_idx=0, chunk=4096, render_step_size=2 / 128)
infer(3); torch.cuda.synchronize()
tr.profile_enable(10); infer(10); torch.cuda.synchronize()
p = tr.profile_read(); inf_ms = p["fwd_chain_camera"][0]
for _ in range(3): tr.step(rays, img, rgbs, 0)
tr.profile_enable(10)
for _ in range(10): tr.step(rays, img, rgbs, 0)
torch.cuda.synchronize()
p = tr.profile_read()
print(f"{os.path.basename(os.environ.get('EONERF_LIB', 'shipped')):28s} infer fwd {inf_ms:.3f} ms | train fwd {p['fwd_chain_camera'][0]:.3f}  bwd {p['bwd_chain_camera'][0]:.3f}  wgrad {p['wgrad_gemm'][0]:.3f}")
