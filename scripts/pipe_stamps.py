"""Diagnostics: per-stage cycle sums of the layer-pipelined trunk backward (EONERF_PIPE_STAMPS=1)."""
import ctypes as C
import os
import sys

os.environ["EONERF_PIPE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eonerf_code_amd import _lib
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.trainer import FusedTrainer

R, N_IMG = 4096, 19
f = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16").cuda()
tr = FusedTrainer(f, lr=5e-4, max_rays=R)
rays, img, rgbs = (t.cuda() for t in synthetic_batch(R, N_IMG))
for _ in range(5):
    tr.step(rays, img, rgbs, 0)
tr.check_device_status()
L = _lib.lib()
L.eonerf_debug_pipe_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(80000, dtype=np.uint64)
n = L.eonerf_debug_pipe_stamps(tr.ctx, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf[:n].reshape(-1, 7, 8, 16).astype(np.float64)
print("pipelines", st.shape[0], "steps/pipeline", st[:, 0, 0, 6].mean())
print("stage | wave | total cyc/step | slow-path | counted wait | barrier | dX | epilogue+stores | DMA issue | tr reads | dW | slow steps %")
for s in range(7):
    for w in range(8):
        v = st[:, s, w]
        nk = v[:, 6].mean()
        c = lambda i: v[:, i].mean() / nk
        print(f"  {s} L{7 - s} | w{w} | {c(0):7.0f} | {c(1):7.0f} | {c(2):6.0f} | {c(3):7.0f} | {c(4):6.0f} | {c(7):6.0f} | {c(8):6.0f} | {c(9):6.0f} | {c(10):6.0f} | {100 * v[:, 5].mean() / nk:5.1f}")
