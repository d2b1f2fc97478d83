"""Runs a few inference renders and training steps so that one rocprofv3 run sees every kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.trainer import FusedTrainer
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda")
torch.manual_seed(42)
f = EONerfMLP(19, radiometric_normalization=True, precision=prec).to(dev)
rays, img, rgbs = (t.to(dev) for t in synthetic_batch(4096, 19))
with torch.no_grad():
    for _ in range(steps):
        render_image(f, None, define_satrays_from_tensors(rays, img[:, None]), None, None, epoch_idx=epoch, chunk=4096, render_step_size=2/128)
tr = FusedTrainer(f)
for _ in range(steps):
    tr.step(rays, img, rgbs, epoch)
torch.cuda.synchronize()
print("done")
