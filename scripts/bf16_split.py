"""Diagnostic (GPU box): where does the per-ray altitude shift of the bf16 path on a TRAINED field come from?  Trains the bf16 field on the
synthetic terrain (tests/bf16_common.py), then evaluates the CPU oracle on the trained weights with bf16 rounding switched on selectively:
weights vs activations, and per layer group.  Prints mean / p99 of |alt_variant - alt_fp32| in cm at Z_scale = 50 m."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("EONERF_DETERMINISTIC", "1")
from bf16_common import R, N_IMG, STEP, Z_SCALE, make_fields, train_on_terrain, terrain_batch   # noqa: E402
from oracle import eonerf_oracle as orc   # noqa: E402


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class SelField(orc.Field):
    """round_w: prefixes of the Linear layers whose WEIGHTS are rounded to bf16; round_x: prefixes whose INPUTS are."""

    def __init__(self, sd, round_w=(), round_x=()):
        super().__init__(sd, False)
        self.rw, self.rx = tuple(round_w), tuple(round_x)

    def _lin(self, x, prefix, emulate=None):
        W, b = self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]
        if emulate is False:
            return torch.nn.functional.linear(x, W, b)
        if prefix.startswith(self.rw) and self.rw:
            W = bf16(W)
        if prefix.startswith(self.rx) and self.rx:
            x = bf16(x)
        return x @ W.t() + b


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    f16, f32 = make_fields(seed=42)
    train_on_terrain(f16, 400)
    sd = {k: v.detach().cpu() for k, v in f16.state_dict().items()}
    rays, img, rgb, depth_gt = (t.cpu() for t in terrain_batch(R, seed=901))
    rays, img = rays[:n], img[:n]
    g = torch.Generator().manual_seed(1)
    u_cam, u_sun = torch.rand(n, 128, generator=g), torch.rand(n, 128, generator=g)
    sr = orc.define_satrays_from_tensors(rays, img[:, None])
    TRUNK = tuple(f"base_mlp.hidden_layers.{i}" for i in range(8))
    ALL = ("base_mlp", "sigma_layer", "bottleneck_layer", "albedo_mlp", "transient_mlp", "transient_scalar", "transient_beta")
    variants = {
        "fp32": ((), ()),
        "all (weights + activations)": (ALL, ALL),
        "weights only, all layers": (ALL, ()),
        "activations only, all layers": ((), ALL),
        "weights: layer 0": (TRUNK[:1], ()),
        "weights: layers 1-4": (TRUNK[1:5], ()),
        "weights: layer 5": (TRUNK[5:6], ()),
        "weights: layers 6-7": (TRUNK[6:], ()),
        "weights: sigma row": (("sigma_layer",), ()),
        "weights: layer 0 + sigma": (TRUNK[:1] + ("sigma_layer",), ()),
        "activations: into layer 0 (encoding)": ((), TRUNK[:1]),
        "activations: into layers 1-7": ((), TRUNK[1:]),
        "activations: into sigma": ((), ("sigma_layer",)),
        "everything except layer-0 weights": (tuple(p for p in ALL if p != "base_mlp") + TRUNK[1:], ALL),
    }
    alts = {}
    with torch.no_grad():
        for name, (rw, rx) in variants.items():
            out, _ = orc.render_rays(SelField(sd, rw, rx), sr, u_cam, u_sun, 3, STEP)
            alts[name] = orc.altitude_from_depth(rays, out[:, 3:4], Z_SCALE, 20.0)
    ref = alts["fp32"]
    res = {}
    for name, a in alts.items():
        d = (a - ref).abs() * 100
        res[name] = {"mean_cm": round(d.mean().item(), 3), "p99_cm": round(d.quantile(0.99).item(), 3), "bias_cm": round(((a - ref) * 100).mean().item(), 3)}
        print(f"{name:45s} mean {res[name]['mean_cm']:7.3f} cm   p99 {res[name]['p99_cm']:7.3f} cm   signed mean {res[name]['bias_cm']:7.3f} cm", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
