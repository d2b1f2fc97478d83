"""Diagnostic (EO_STAMP build, scripts/stamp.sh): share of the chain kernels' wave time spent in the copy wait and at the chunk barrier."""
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
out = (C.c_ulonglong * 8)()
def rd(tag, which):
    torch.cuda.synchronize(); getattr(L, "eonerf_debug_read_" + which)(out)
    tot, w, b, fl, n = out[0], out[1], out[2], out[3], out[4]
    if n: print(f"{tag}: waves={n} cycles/wave={tot/n:.0f} copy-wait={100*w/tot:.1f}% barrier-wait={100*b/tot:.1f}% slab-flush={100*fl/tot:.1f}%")
sr = define_satrays_from_tensors(rays, img[:, None])
def infer(n):
    with torch.no_grad():
        for _ in range(n): render_image(f, None, sr, None, None, epoch_idx=0, chunk=4096, render_step_size=2 / 128)
infer(3); rd("warm", "fwd")
infer(5); rd("fwd inference", "fwd")
tr = FusedTrainer(f)
for _ in range(3): tr.step(rays, img, rgbs, 0)
rd("warm", "fwd"); rd("warm", "bwd")
for _ in range(5): tr.step(rays, img, rgbs, 0)
rd("fwd train (rgb state)", "fwd"); rd("bwd heads chain (rgb state)", "bwd")
for _ in range(3): tr.step(rays, img, rgbs, 3)
rd("warm", "fwd"); rd("warm", "bwd")
for _ in range(5): tr.step(rays, img, rgbs, 3)
rd("fwd train (full state: camera + sun launches)", "fwd"); rd("bwd heads chains (full state: camera + sun)", "bwd")
