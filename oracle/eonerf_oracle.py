"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the EO-NeRF per-ray hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (eonerf_code_amd/) never does and fails loudly
when the HIP library is missing.

Every function cites the reference file:line (relative to /root/reference) it
restates.  Differences from the reference are *interface* only: the jitter noise
`u` is an explicit input (the reference draws torch.rand_like inside
perturb_z_vals, sat_rendering.py:52) and weights come in as a state_dict with the
reference's 44 keys.

PARITY STATUS
  pinned by golden vectors captured from the reference's own code in this
  container (tests/golden/make_golden.py -> tests/golden/*.npz):
    encoder (G1), MLP skip-concat (G2), EONerfMLP.forward/query_density (G3),
    satnerf_sampling/perturb_z_vals (G4), weights_from_sigma cross-check (G5),
    metrics.* (G6), autograd grads of G3 (G7), and -- with nerfacc restated --
    EONerfMLP.rendering / compute_geometric_shadows / render_image (G8).
  UNPINNED: the nerfacc v0.5.2 arithmetic itself (oracle/nerfacc_restated.py)
  and RPC ray generation through rpcm/pyproj (both un-vendored third-party).
"""
import math
from collections import namedtuple

import torch
import torch.nn.functional as F

from . import nerfacc_restated as _nv

SatRays = namedtuple("Rays", ("origins", "viewdirs", "sundirs", "img_idx", "t_near", "t_far"))

POS_L = 10   # radiance_fields/eonerf.py:80
VIEW_L = 4   # radiance_fields/eonerf.py:81
BETA_MIN = 0.05  # radiance_fields/eonerf.py:87


def define_satrays_from_tensors(rays, ts):
    """datasets/satellite.py:23-26."""
    return SatRays(origins=rays[:, :3], viewdirs=rays[:, 3:6], sundirs=rays[:, 8:11],
                   img_idx=ts, t_near=rays[:, 6:7], t_far=rays[:, 7:8])


# --------------------------------------------------------------------------- encoder / MLP
def sinusoidal_encode(x: torch.Tensor, L: int) -> torch.Tensor:
    """radiance_fields/mlp.py:190-208 with min_deg=0, use_identity=True.

    latent = [x, sin(2^k x) (freq-major, xyz-minor), sin(2^k x + fp32(pi/2))]; the
    trailing all-ones frequency mask (mlp.py:207) is a numerical no-op.
    """
    scales = torch.tensor([2 ** i for i in range(L)], device=x.device)  # int64 buffer, mlp.py:177-179
    xb = torch.reshape(x[..., None, :] * scales[:, None], list(x.shape[:-1]) + [L * x.shape[-1]])
    latent = torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))
    return torch.cat([x, latent], dim=-1)


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class Field:
    """EONerfMLP restated over a plain state_dict (radiance_fields/eonerf.py:69-170).

    emulate_bf16=True rounds every Linear's input and weight to bf16 (fp32 accumulate,
    fp32 bias) -- the arithmetic model of the HIP bf16 MFMA path; the tiny per-ray
    ambient head stays fp32 there as it does in the kernel.
    """

    def __init__(self, sd, emulate_bf16=False):
        self.sd = sd
        self.bf16 = emulate_bf16
        self.net_depth = sum(1 for k in sd if k.startswith("base_mlp.hidden_layers.") and k.endswith(".weight"))
        self.skip = 4
        self.has_radiometric = "radiometricT_enc.weight" in sd

    def _lin(self, x, prefix, emulate=None):
        W, b = self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]
        if self.bf16 if emulate is None else emulate:
            return _bf16(x) @ _bf16(W).t() + b
        return F.linear(x, W, b)

    def _mlp(self, x, prefix, depth, skip, emulate=None):
        """radiance_fields/mlp.py:87-97: Linear -> ReLU, concat [x, inputs] AFTER layer i when i%skip==0 and i>0."""
        inputs = x
        for i in range(depth):
            x = torch.relu(self._lin(x, f"{prefix}.hidden_layers.{i}", emulate))
            if skip is not None and i % skip == 0 and i > 0:
                x = torch.cat([x, inputs], dim=-1)
        return x

    def trunk(self, x):
        return self._mlp(sinusoidal_encode(x, POS_L), "base_mlp", self.net_depth, self.skip)

    def query_density(self, x):
        """radiance_fields/eonerf.py:141-145 (Softplus: beta=1, threshold=20)."""
        return F.softplus(self._lin(self.trunk(x), "sigma_layer.output_layer"))

    def query_opacity(self, x, step_size):
        """radiance_fields/eonerf.py:147-152."""
        return self.query_density(x) * step_size

    def ambient(self, sun_dirs):
        """ambient head, radiance_fields/eonerf.py:163-164,132-139 (always fp32, see class doc)."""
        h = self._mlp(sinusoidal_encode(sun_dirs, VIEW_L), "ambient_mlp", 1, None, emulate=False)
        return torch.sigmoid(self._lin(h, "ambient_mlp.output_layer", emulate=False))

    def forward(self, x, sun_dirs, img_indices):
        """radiance_fields/eonerf.py:154-170."""
        h = self.trunk(x)
        sigma = F.softplus(self._lin(h, "sigma_layer.output_layer"))
        emb = self.sd["transient_encoder.weight"][img_indices.reshape(-1)]
        ambient = self.ambient(sun_dirs)
        if self.bf16:
            return (sigma,) + self._heads_bf16_model(h, emb, ambient)
        bott = self._lin(h, "bottleneck_layer.output_layer")
        a = self._mlp(bott, "albedo_mlp", 1, None)
        albedo = torch.sigmoid(self._lin(a, "albedo_mlp.output_layer"))
        t = self._mlp(torch.cat([bott, emb], dim=-1), "transient_mlp", 4, None)
        ts = torch.sigmoid(self._lin(t, "transient_scalar.output_layer"))
        tb = F.softplus(self._lin(t, "transient_beta.output_layer"))
        return sigma, albedo, ambient, ts, tb


def _heads_bf16_model(self, h, emb, ambient):
    """Arithmetic model of the HIP bf16 kernels behind the trunk (NOT a reference function: the fp32 path above is the reference's
    graph, radiance_fields/eonerf.py:157-168).  The bottleneck layer has an identity activation (eonerf.py:108-113), so the kernels
    compose it with the two head layers that read it -- W_f = W_head W_bott and b_f = W_head b_bott + b_head, formed in fp32 from the
    fp32 master weights -- and round the product ONCE to bf16 (csrc/eonerf_pack.h); the bottleneck output itself is never formed."""
    sd = self.sd
    Wb, bb = sd["bottleneck_layer.output_layer.weight"], sd["bottleneck_layer.output_layer.bias"]
    Wa, ba = sd["albedo_mlp.hidden_layers.0.weight"], sd["albedo_mlp.hidden_layers.0.bias"]
    Wt, bt = sd["transient_mlp.hidden_layers.0.weight"], sd["transient_mlp.hidden_layers.0.bias"]
    a = torch.relu(_bf16(h) @ _bf16(Wa @ Wb).t() + (Wa @ bb + ba))
    albedo = torch.sigmoid(self._lin(a, "albedo_mlp.output_layer"))
    t = torch.relu(_bf16(h) @ _bf16(Wt[:, :256] @ Wb).t() + _bf16(emb) @ _bf16(Wt[:, 256:]).t() + (Wt[:, :256] @ bb + bt))
    for i in range(1, 4):
        t = torch.relu(self._lin(t, f"transient_mlp.hidden_layers.{i}"))
    ts = torch.sigmoid(self._lin(t, "transient_scalar.output_layer"))
    tb = F.softplus(self._lin(t, "transient_beta.output_layer"))
    return albedo, ambient, ts, tb


Field._heads_bf16_model = _heads_bf16_model


# --------------------------------------------------------------------------- sampler
def perturb_z_vals(z_vals, u):
    """sat_rendering.py:46-54 with perturb_rand := u (the reference draws rand_like(z_vals))."""
    mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
    upper = torch.cat([mid, z_vals[:, -1:]], -1)
    lower = torch.cat([z_vals[:, :1], mid], -1)
    return lower + (upper - lower) * u


def satnerf_sampling(origins, viewdirs, u, render_step_size, near=None):
    """sat_rendering.py:56-84.  far is always near+2 (the t_far column is ignored, :254,258)."""
    if near is None:
        near = torch.zeros_like(origins[:, 0:1])
    far = near + 2
    n_samples = int(2 / render_step_size)
    z_steps = torch.linspace(0, 1, n_samples, device=origins.device)
    z_vals = near * (1 - z_steps) + far * z_steps
    z_vals = perturb_z_vals(z_vals, u)
    n_rays = origins.shape[0]
    t_ends = (z_vals[:, :-1] + (z_vals[:, 1:] - z_vals[:, :-1])).flatten()
    t_starts = z_vals[:, :-1].flatten()
    ray_indices = torch.arange(n_rays, device=origins.device).repeat_interleave(n_samples - 1)
    mids = (t_starts + t_ends)[:, None] / 2.0
    xyz = origins[ray_indices] + viewdirs[ray_indices] * mids
    mask = torch.sum(torch.abs(xyz) >= 1, dim=1) == 0   # filter_pts_outside_cube, sat_rendering.py:18-22
    return ray_indices[mask], t_starts[mask], t_ends[mask]


def count_pts_per_ray(n_rays, ray_indices):
    """sat_rendering.py:10-16 (fp32 counts)."""
    return torch.bincount(ray_indices, minlength=n_rays).to(torch.float32)


def _last_idx(ray_indices, n_rays):
    counts = torch.bincount(ray_indices, minlength=n_rays)
    present = counts > 0
    last = torch.cumsum(counts, 0) - 1
    return last[present], present


# --------------------------------------------------------------------------- compositing
def rendering(field, rays, t_starts, t_ends, ray_indices):
    """EONerfMLP.rendering, radiance_fields/eonerf.py:196-248."""
    n_rays = rays.origins.shape[0]
    z_vals = (t_starts + t_ends)[:, None] / 2.0
    positions = rays.origins[ray_indices] + rays.viewdirs[ray_indices] * z_vals
    last, _ = _last_idx(ray_indices, n_rays)
    t_ends = t_ends.clone()
    t_ends[last] = 1e10                                   # eonerf.py:218-220 (after the mid points are taken)
    sigma, albedo, ambient, ts, tb = field.forward(positions, rays.sundirs[ray_indices], rays.img_idx[ray_indices])
    w, _, _ = _nv.render_weight_from_density(t_starts, t_ends, sigma.squeeze(-1), ray_indices, n_rays)
    acc = lambda v: _nv.accumulate_along_rays(w, v, ray_indices, n_rays)
    depth, albedo_r, ambient_r, ts_r, tb_r = acc(z_vals), acc(albedo), acc(ambient), acc(ts), acc(tb)
    tb_r = tb_r + BETA_MIN
    return albedo_r, depth, tb_r, ts_r, ambient_r, torch.ones_like(depth)


def render_depth(field, rays, t_starts, t_ends, ray_indices):
    """EONerfMLP.render_depth, radiance_fields/eonerf.py:172-194."""
    n_rays = rays.origins.shape[0]
    z_vals = (t_starts + t_ends)[:, None] / 2.0
    positions = rays.origins[ray_indices] + rays.viewdirs[ray_indices] * z_vals
    last, _ = _last_idx(ray_indices, n_rays)
    t_ends = t_ends.clone()
    t_ends[last] = 1e10
    sigma = field.query_density(positions).squeeze(-1)
    w, _, _ = _nv.render_weight_from_density(t_starts, t_ends, sigma, ray_indices, n_rays)
    return _nv.accumulate_along_rays(w, z_vals, ray_indices, n_rays)


def compute_geometric_shadows(field, rays, depth, u_sun, render_step_size):
    """sat_rendering.py:87-118 (depth is NOT detached, :90; no 1e10 patch on this pass)."""
    n_rays = rays.origins.shape[0]
    sc_origins = rays.origins + torch.hstack([depth, depth, depth]) * rays.viewdirs
    sc_viewdirs = -1.0 * rays.sundirs
    ri, ts_, te_ = satnerf_sampling(sc_origins, sc_viewdirs, u_sun, render_step_size)
    sc_pts = count_pts_per_ray(n_rays, ri)
    z = (ts_ + te_)[:, None] / 2.0
    pos = sc_origins[ri] + sc_viewdirs[ri] * z
    sigma = field.query_density(pos).squeeze(-1)
    trans, _ = _nv.render_transmittance_from_density(ts_, te_, sigma, ri, n_rays)
    last, present = _last_idx(ri, n_rays)
    geo = torch.ones(n_rays, 1, dtype=depth.dtype, device=depth.device)
    geo = geo.index_put((torch.nonzero(present).squeeze(-1),), trans[last][:, None])
    return geo, sc_pts


def render_rays(field, rays, u_cam, u_sun, epoch_idx, render_step_size, eval=False, u_cam_retry=None):
    """One chunk of render_image, sat_rendering.py:252-312.  Returns ([R,21], n_samples).

    u_cam_retry: noise for the 'any ray empty -> resample' branch (:260-262); near is 0 either way.
    """
    n_rays = rays.origins.shape[0]
    ri, ts_, te_ = satnerf_sampling(rays.origins, rays.viewdirs, u_cam, render_step_size, near=rays.t_near)
    pts = count_pts_per_ray(n_rays, ri)
    if torch.sum(pts == 0) and u_cam_retry is not None:
        ri, ts_, te_ = satnerf_sampling(rays.origins, rays.viewdirs, u_cam_retry, render_step_size)
    albedo, depth, beta, ts, ambient, entropy = rendering(field, rays, ts_, te_, ri)
    ambient = ambient * 0.2                                # :265
    if epoch_idx < 2:                                      # :269-272
        geo = torch.ones(n_rays, 1)
        s = geo
        sc_pts = torch.ones_like(pts)
    else:
        geo, sc_pts = compute_geometric_shadows(field, rays, depth, u_sun, render_step_size)
        s = geo * ts
    opacity_after = torch.ones(n_rays, 2)
    img = (torch.ones(n_rays, dtype=torch.long) * rays.img_idx[0]) if eval else rays.img_idx.reshape(-1)  # :288-291
    rgb = albedo * s + (1 - s) * (ambient * albedo)        # :294
    if field.has_radiometric:
        T = field.sd["radiometricT_enc.weight"][img]
        A, b = T[:, :3], T[:, 3:6]
    else:
        A, b = torch.ones_like(rgb), torch.zeros_like(rgb)
    rgb = torch.clip(A * rgb + b, 0, 1)                    # :304-305
    shadowless = A * albedo + b                            # :306
    out = torch.cat([rgb, depth, albedo, ambient, geo, ts, beta, entropy, pts[:, None], sc_pts[:, None],
                     opacity_after, shadowless], dim=1)     # :311-312
    return out, len(ts_)


RESULT_SLICES = {  # sat_rendering.py:322-334
    "rgb": (0, 3), "depth": (3, 4), "albedo_rgb": (4, 7), "ambient_rgb": (7, 10), "geo_shadows": (10, 11),
    "transient_s": (11, 12), "beta": (12, 13), "entropy": (13, 14), "pts_per_ray": (14, 15),
    "sc_pts_per_ray": (15, 16), "opacity_after_surface": (16, 18), "shadowless_rgb": (18, 21),
}


def weights_from_sigma(z_vals, sigmas):
    """In-tree dense compositing used as a cross-check only, radiance_fields/eonerf.py:37-54."""
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    alphas = 1 - torch.exp(-deltas * torch.relu(sigmas))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    trans = torch.cumprod(shifted, -1)[:, :-1]
    return alphas * trans, trans, alphas


# --------------------------------------------------------------------------- losses (metrics.py)
def uncertainty_aware_loss(gt_rgb, pred_rgb, pred_beta):
    """metrics.py:17-22."""
    color_term = ((pred_rgb - gt_rgb) ** 2 / (2 * pred_beta ** 2)).mean()
    beta_term = (3 + torch.log(pred_beta).mean()) / 2
    return color_term + beta_term, color_term, beta_term


def depth_loss_L2(gt_depth, pred_depth, gt_conf=None, w=100):
    """metrics.py:24-31."""
    valid = gt_depth >= 0
    if gt_conf is not None:
        valid = valid & (gt_conf >= 4)
    return ((pred_depth[valid] - gt_depth[valid]) ** 2).mean() * w


def shadow_loss_L2(smask, geo_shadows):
    """metrics.py:36-58 (the differentiable term only)."""
    diff = (smask <= 0.5) * (geo_shadows - smask) ** 2
    mean_diff = torch.sum(diff) / (torch.sum(smask <= 0.5) + 1e-6)
    pct = torch.sum(smask <= 0.5) / torch.sum(smask >= 0)
    return pct * mean_diff


def psnr(pred, gt):
    """metrics.py:60-69."""
    return -10 * torch.log10(torch.mean((pred - gt) ** 2))


def train_loss(out, pixels, epoch_idx):
    """train_eonerf.py:139-143."""
    if epoch_idx < 2:
        return F.mse_loss(out[:, 0:3], pixels)
    return uncertainty_aware_loss(pixels, out[:, 0:3], out[:, 12:13])[0]


# --------------------------------------------------------------------------- altitude (1 cm criterion)
def altitude_from_depth(rays, depth, z_scale, z_offset):
    """datasets/satellite.py:502-533, UTM branch, Z only, fp64."""
    r, d = rays.double(), depth.double()
    return (r[:, 2] + r[:, 5] * d.view(-1)) * z_scale + z_offset


# --------------------------------------------------------------------------- weights / synthetic inputs
LAYER_SHAPES = [  # (key prefix, out, in) -- SURVEY.md 8b state_dict manifest
    ("base_mlp.hidden_layers.0", 256, 63), ("base_mlp.hidden_layers.1", 256, 256),
    ("base_mlp.hidden_layers.2", 256, 256), ("base_mlp.hidden_layers.3", 256, 256),
    ("base_mlp.hidden_layers.4", 256, 256), ("base_mlp.hidden_layers.5", 256, 319),
    ("base_mlp.hidden_layers.6", 256, 256), ("base_mlp.hidden_layers.7", 256, 256),
    ("sigma_layer.output_layer", 1, 256), ("bottleneck_layer.output_layer", 256, 256),
    ("albedo_mlp.hidden_layers.0", 128, 256), ("albedo_mlp.output_layer", 3, 128),
    ("transient_mlp.hidden_layers.0", 128, 260), ("transient_mlp.hidden_layers.1", 128, 128),
    ("transient_mlp.hidden_layers.2", 128, 128), ("transient_mlp.hidden_layers.3", 128, 128),
    ("transient_scalar.output_layer", 1, 128), ("transient_beta.output_layer", 1, 128),
    ("ambient_mlp.hidden_layers.0", 128, 27), ("ambient_mlp.output_layer", 3, 128),
]


def random_state_dict(n_img, seed=42, bias_scale=0.0, radiometric_jitter=0.0):
    """Xavier-uniform weights (radiance_fields/mlp.py:22-28,67-85), zero (or small random) bias,
    N(0,1) transient embedding (nn.Embedding default), radiometric init [1,1,1,0,...] (eonerf.py:92-94).
    Not bit-identical to the reference's construction-order RNG stream; parity tests load the same
    dict into both sides so this does not matter."""
    g = torch.Generator().manual_seed(seed)
    sd = {"posi_encoder.scales": torch.tensor([2 ** i for i in range(POS_L)]),
          "view_encoder.scales": torch.tensor([2 ** i for i in range(VIEW_L)]),
          "transient_encoder.weight": torch.randn(n_img, 4, generator=g)}
    rad = torch.cat([torch.ones(n_img, 3), torch.zeros(n_img, 6)], 1)
    if radiometric_jitter:
        rad = rad + radiometric_jitter * torch.randn(n_img, 9, generator=g)
    sd["radiometricT_enc.weight"] = rad
    for name, o, i in LAYER_SHAPES:
        bound = math.sqrt(6.0 / (i + o))
        sd[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = bias_scale * (torch.rand(o, generator=g) * 2 - 1)
    return sd


def closed_form_state_dict(n_img):
    """Deterministic filler weights used by golden G3/G7/G8: w[i,j] = s*sin(0.37*(i*in+j)+c), s = Xavier bound.
    Only this formula and the outputs are committed; both sides regenerate the weights from it."""
    sd = {"posi_encoder.scales": torch.tensor([2 ** i for i in range(POS_L)]),
          "view_encoder.scales": torch.tensor([2 ** i for i in range(VIEW_L)])}
    ii = torch.arange(n_img, dtype=torch.float64)[:, None]
    sd["transient_encoder.weight"] = torch.sin(1.3 * ii + 0.7 * torch.arange(4, dtype=torch.float64)[None]).float()
    jj = torch.arange(9, dtype=torch.float64)[None]
    base = torch.cat([torch.ones(n_img, 3), torch.zeros(n_img, 6)], 1).double()
    sd["radiometricT_enc.weight"] = (base + 0.05 * torch.sin(0.9 * ii + 1.1 * jj)).float()
    for c, (name, o, i) in enumerate(LAYER_SHAPES):
        idx = torch.arange(o * i, dtype=torch.float64).reshape(o, i)
        s = math.sqrt(6.0 / (i + o))
        sd[name + ".weight"] = (s * torch.sin(0.37 * idx + c)).float()
        sd[name + ".bias"] = (0.01 * torch.sin(0.11 * torch.arange(o, dtype=torch.float64) + c)).float()
    return sd


def get_dir_vec_from_el_az(elevation_deg, azimuth_deg):
    """datasets/satellite.py:57-63 (float64 numpy in the reference)."""
    el = math.radians(90 - elevation_deg)
    az = math.radians(azimuth_deg)
    return [-math.sin(az) * math.cos(el), -math.cos(az) * math.cos(el), -math.sin(el)]


def synthetic_batch(n_rays, n_img, seed=1234, n_samples=128):
    """SURVEY.md 8d synthetic JAX_068-like batch: rays[R,11], ts[R,1], rgbs[R,3], u_cam, u_sun [R,128]."""
    g = torch.Generator().manual_seed(seed)
    o = torch.empty(n_rays, 3)
    o[:, :2] = torch.rand(n_rays, 2, generator=g) * 1.8 - 0.9
    o[:, 2] = 0.98
    d = torch.cat([0.15 * torch.randn(n_rays, 2, generator=g), -torch.ones(n_rays, 1)], 1)
    d = d / d.norm(dim=1, keepdim=True)
    el = torch.rand(n_img, generator=g) * 40 + 30
    az = torch.rand(n_img, generator=g) * 180 + 90
    # sun_elevation -> get_sun_dirs(90 - elev, az) -> get_dir_vec_from_el_az (datasets/satellite.py:457,486-500)
    sun = torch.tensor([get_dir_vec_from_el_az(90 - float(e), float(a)) for e, a in zip(el, az)], dtype=torch.float32)
    sun = sun / sun.norm(dim=1, keepdim=True)
    ts = torch.randint(0, n_img, (n_rays, 1), generator=g)
    rays = torch.cat([o, d, torch.zeros(n_rays, 1), 2 * torch.ones(n_rays, 1), sun[ts[:, 0]]], 1)
    rgbs = torch.rand(n_rays, 3, generator=g)
    u_cam = torch.rand(n_rays, n_samples, generator=g)
    u_sun = torch.rand(n_rays, n_samples, generator=g)
    return rays, ts, rgbs, u_cam, u_sun


def train_step(sd_params, rays, ts, rgbs, u_cam, u_sun, epoch_idx, render_step_size, opt=None, emulate_bf16=False):
    """train_eonerf.py:104-161 for one batch: render -> loss -> backward -> Adam.  sd_params: dict whose float
    tensors have requires_grad=True.  Returns (loss, out)."""
    field = Field(sd_params, emulate_bf16)
    satrays = define_satrays_from_tensors(rays, ts)
    out, _ = render_rays(field, satrays, u_cam, u_sun, epoch_idx, render_step_size)
    loss = train_loss(out, rgbs, epoch_idx)
    if opt is not None:
        opt.zero_grad()
    loss.backward()
    if opt is not None:
        opt.step()
    return loss.detach(), out.detach()
