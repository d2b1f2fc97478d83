import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eonerf_code_amd import _lib
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.trainer import FusedTrainer
from eonerf_code_amd.sat_rendering import render_image
from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
dev = torch.device("cuda")
f = EONerfMLP(19, radiometric_normalization=True).to(dev)
rays, img, rgbs = (t.to(dev) for t in synthetic_batch(4096, 19))
L = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * 4)()
def rd(tag):
    torch.cuda.synchronize(); L.eonerf_debug_read(out)
    tot, w, b, n = out[0], out[1], out[2], out[3]
    if n: print(f"{tag}: waves={n} avg kernel cycles/wave={tot/n:.0f} vmcnt-wait={100*w/tot:.1f}% barrier-wait={100*b/tot:.1f}%")
with torch.no_grad():
    for _ in range(3): render_image(f, None, define_satrays_from_tensors(rays, img[:, None]), None, None, epoch_idx=0, chunk=4096, render_step_size=2/128)
rd("warm")
with torch.no_grad():
    for _ in range(3): render_image(f, None, define_satrays_from_tensors(rays, img[:, None]), None, None, epoch_idx=0, chunk=4096, render_step_size=2/128)
rd("fwd inference (full)")
tr = FusedTrainer(f)
for _ in range(2): tr.step(rays, img, rgbs, 0)
rd("warm train")
for _ in range(3): tr.step(rays, img, rgbs, 0)
rd("fwd train mode 2 (bwd kernel not stamped)")
