"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the vanilla NeRF field of BASELINE.json configs[0]
("nerf_synthetic lego, vanilla 8-layer MLP, 256 rays x 64 samples, CPU fp32 via train_mlp_nerf.py: plumbing, no GPU").

Restates radiance_fields/mlp.py:114-165 (NerfMLP) and :211-250 (VanillaNeRFRadianceField) of /root/reference over a plain state_dict,
on top of the encoder / MLP restatement of oracle/eonerf_oracle.py (mlp.py:87-101,190-208).  train_mlp_nerf.py itself is broken as
shipped (imports a non-existent `utils2`), so what can be pinned is the field and its autograd.

PARITY STATUS: pinned by golden G9 (tests/golden/g9_vanilla.npz), produced by the reference's own VanillaNeRFRadianceField in this
container (tests/golden/make_golden.py: closed-form weights and inputs, regenerated on both sides from the formulas below).
"""
import torch
import torch.nn.functional as F

from .eonerf_oracle import sinusoidal_encode

POS_L, VIEW_L = 10, 4      # mlp.py:222-223


def fill(name, shape):
    """Closed-form weights of golden G9 (same formula as tests/golden/make_golden.py::vanilla_fill)."""
    c = sum(ord(ch) for ch in name) % 17
    n = 1
    for d in shape:
        n *= d
    idx = torch.arange(n, dtype=torch.float64).reshape(shape)
    fan = shape[-1] + shape[0] if len(shape) == 2 else 1
    scale = (6.0 / fan) ** 0.5 if len(shape) == 2 else 0.01
    return (scale * torch.sin(0.37 * idx + c)).float()


def inputs(n_rays=256, n_samples=64):
    """configs[0] batch: positions [R,S,3], one view direction per ray [R,3] (same formula as make_golden.py::vanilla_inputs)."""
    r = torch.arange(n_rays, dtype=torch.float64)[:, None]
    k = torch.arange(n_samples, dtype=torch.float64)[None, :]
    x = torch.stack([torch.sin(0.11 * r + 0.07 * k), torch.cos(0.05 * r - 0.13 * k), torch.sin(0.017 * r * k + 0.3)], dim=-1).float() * 0.95
    d = torch.stack([torch.sin(0.3 * r[:, 0]), torch.cos(0.2 * r[:, 0]), -torch.ones(n_rays, dtype=torch.float64)], dim=-1)
    d = (d / d.norm(dim=-1, keepdim=True)).float()
    return x, d


def state_dict_from_manifest(manifest):
    """manifest entries "key:(shape):dtype" (the reference module's state_dict) -> closed-form state_dict."""
    sd = {}
    for entry in manifest:
        key, shape, dtype = str(entry).split(":")
        shape = tuple(int(v) for v in shape.strip("()").split(",") if v.strip())
        if dtype.startswith("int"):
            sd[key] = torch.tensor([2 ** i for i in range(shape[0])])          # SinusoidalEncoder.scales, mlp.py:177-179
        else:
            sd[key] = fill(key, shape)
    return sd


class VanillaField:
    """VanillaNeRFRadianceField over a state_dict: 8 x 256 trunk with skip-concat after layer 4 (mlp.py:87-97), raw sigma from the
    trunk, bottleneck (identity) + encoded view direction -> 128 ReLU -> 3 (mlp.py:133-163), sigmoid / relu on top (:247-250)."""

    def __init__(self, sd):
        self.sd = sd
        self.depth = sum(1 for k in sd if k.startswith("mlp.base.hidden_layers.") and k.endswith(".weight"))
        self.depth_c = sum(1 for k in sd if k.startswith("mlp.rgb_layer.hidden_layers.") and k.endswith(".weight"))

    def _lin(self, x, prefix):
        return F.linear(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"])

    def _trunk(self, x):
        inputs = x
        for i in range(self.depth):
            x = torch.relu(self._lin(x, f"mlp.base.hidden_layers.{i}"))
            if i % 4 == 0 and i > 0:                                  # skip_layer = 4: concat AFTER the activation, x first
                x = torch.cat([x, inputs], dim=-1)
        return x

    def query_density(self, x):
        """mlp.py:240-243."""
        h = self._trunk(sinusoidal_encode(x, POS_L))
        return torch.relu(self._lin(h, "mlp.sigma_layer.output_layer"))

    def query_opacity(self, x, step_size):
        """mlp.py:233-238."""
        return self.query_density(x) * step_size

    def forward(self, x, condition):
        """mlp.py:245-250 with a per-ray condition [R,3] expanded over the samples of x [R,S,3] (:154-159)."""
        h = self._trunk(sinusoidal_encode(x, POS_L))
        raw_sigma = self._lin(h, "mlp.sigma_layer.output_layer")
        c = sinusoidal_encode(condition, VIEW_L)
        if c.shape[:-1] != h.shape[:-1]:
            c = c.view([c.shape[0]] + [1] * (h.dim() - c.dim()) + [c.shape[-1]]).expand(list(h.shape[:-1]) + [c.shape[-1]])
        y = torch.cat([self._lin(h, "mlp.bottleneck_layer.output_layer"), c], dim=-1)
        for i in range(self.depth_c):
            y = torch.relu(self._lin(y, f"mlp.rgb_layer.hidden_layers.{i}"))
        raw_rgb = self._lin(y, "mlp.rgb_layer.output_layer")
        return torch.sigmoid(raw_rgb), torch.relu(raw_sigma)
