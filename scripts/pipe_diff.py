"""Diagnostics: per-tensor difference between the pipelined trunk backward and the chain + GEMM path (same batch, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_bwd_pipe import _field, _grads
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
f_old, f_new = _field(False), _field(True)
l0, g0, _ = _grads(f_old, R, 0)
l1, g1, _ = _grads(f_new, R, 0)
for (name, p), a, b in zip(f_old.named_parameters(), f_old.grad_views(g0), f_new.grad_views(g1)):
    rel = (a - b).norm().item() / (a.norm().item() + 1e-30)
    print(f"{name:45s} rel {rel:9.2e}  |a| {a.norm().item():9.3e}")
    if name == "base_mlp.hidden_layers.0.weight" or name == "base_mlp.hidden_layers.5.weight":
        d = (a - b)
        print("   per-row-block rel:", [round((d[32 * i:32 * i + 32].norm() / (a[32 * i:32 * i + 32].norm() + 1e-30)).item(), 3) for i in range(8)])
        if name.endswith("5.weight"):
            print("   cols <256:", (d[:, :256].norm() / a[:, :256].norm()).item(), " skip cols:", (d[:, 256:].norm() / a[:, 256:].norm()).item())
