"""render_image -- host-side mirror of sat_rendering.py:176-335 over libeonerf_hip.so.

Same signature, same result dict (12 keys, sat_rendering.py:322-334) and the same chunk loop semantics; per chunk ONE
library call replaces satnerf_sampling + EONerfMLP.rendering + compute_geometric_shadows + the irradiance /
radiometric model (sat_rendering.py:252-312), and autograd is one library call back.

The jitter noise the reference draws with torch.rand_like inside perturb_z_vals (:52) is drawn INSIDE the sampler
kernels (Philox4x32-10, the generator family behind torch.rand; U[0,1) with 24 random bits) or can be injected through
`noise=` for parity tests.
"""
import ctypes as C

import torch

from . import _lib
from .datasets.satellite import SatRays, namedtuple_map, satrays_to_table
from .radiance_fields.eonerf import _ptr, _stream

RESULT_SLICES = (("rgb", 0, 3), ("depth", 3, 4), ("albedo_rgb", 4, 7), ("ambient_rgb", 7, 10), ("geo_shadows", 10, 11),
                 ("transient_s", 11, 12), ("beta", 12, 13), ("entropy", 13, 14), ("pts_per_ray", 14, 15),
                 ("sc_pts_per_ray", 15, 16), ("opacity_after_surface", 16, 18), ("shadowless_rgb", 18, 21))

_ZSTEPS = {}


def _zsteps(device, n_samples=128):
    z = _ZSTEPS.get((device, n_samples))
    if z is None:
        z = torch.linspace(0, 1, n_samples).to(device)     # the reference's fp32 table (sat_rendering.py:67), computed once per size
        _ZSTEPS[(device, n_samples)] = z
    return z


def n_samples_of(render_step_size):
    """sat_rendering.py:64: n_samples = int(2 / render_step_size); a ray's samples live in one wavefront (64 lanes x up to 4 slots)."""
    n = int(2 / render_step_size)
    if not 2 <= n <= 256:
        raise ValueError(f"render_step_size={render_step_size} gives {n} samples/ray; the HIP path supports 2 .. 256 (128: run_JAX_RGB.sh:11)")
    return n


def count_number_of_pts_per_nerfacc_ray(rays, ray_indices):
    """sat_rendering.py:10-16: fp32 number of samples of every ray (0 for rays without samples)."""
    n_rays = rays.origins.shape[0]
    return torch.bincount(ray_indices, minlength=n_rays).to(rays.origins.dtype)


@torch.no_grad()
def satnerf_sampling(origins, viewdirs, sampling_args, near=None, far=None, perturb=True, noise=None, radiance_field=None):
    """sat_rendering.py:56-84: (ray_indices, t_starts, t_ends) of the cube-filtered stratified samples.
    `far` is ignored exactly as in the reference (far = near + 2); noise [R,n_samples] replaces the rand_like draw."""
    ns = n_samples_of(sampling_args["render_step_size"])
    n, dev = origins.shape[0], origins.device
    table = torch.zeros(n, 11, dtype=torch.float32, device=dev)
    table[:, 0:3], table[:, 3:6] = origins, viewdirs
    if near is not None:
        table[:, 6:7] = near.reshape(n, 1)
    # noise=None: the jitter is drawn inside the sampler kernel (Philox); perturb=False: no jitter at all (:70-71)
    u = None if (noise is None or not perturb) else noise.to(dev, torch.float32).contiguous()
    cap = max(n * (ns - 1), 1)
    ri = torch.empty(cap, dtype=torch.int64, device=dev)
    ts_, te_ = torch.empty(cap, dtype=torch.float32, device=dev), torch.empty(cap, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    field = radiance_field if radiance_field is not None else _any_field(dev)
    field._context()
    field.set_n_samples(ns)
    nb = L.eonerf_render_workspace_bytes(field._ctx, n, _lib.F_ONLY_DEPTH)
    ws = field._workspace("render", nb)
    _lib.check(L.eonerf_sample_rays(field._ctx, _ptr(table), _ptr(_zsteps(dev, ns)), _ptr(u), 1 if perturb else 0, n, _ptr(ri), _ptr(ts_), _ptr(te_), None,
                                    _ptr(cnt), _ptr(ws), ws.numel(), _stream()))
    k = int(cnt.item())
    return ri[:k], ts_[:k], te_[:k]


_FIELD_FOR_SAMPLING = {}


def _any_field(dev):
    """The sampler needs a library context (workspace carving) but no weights: keep one tiny module per device."""
    f = _FIELD_FOR_SAMPLING.get(dev)
    if f is None:
        from .radiance_fields.eonerf import EONerfMLP
        f = EONerfMLP(1).to(dev)
        f._ensure_packed()
        _FIELD_FOR_SAMPLING[dev] = f
    return f


class _RenderChunk(torch.autograd.Function):
    """One chunk of render_image as a differentiable op; the parameters are inputs so autograd routes their grads."""

    @staticmethod
    def forward(ctx, field, table, img, flags, export, u_cam, u_retry, u_sun, *params):
        L = _lib.lib()
        train = bool(flags & _lib.F_TRAIN)
        native, flat = field._native(export and not train)      # export renders of a bf16 field: its fp32 context (EONerfMLP.eval_precision)
        ns = field._n_samples                                   # (render_image has set it from its render_step_size)
        n = table.shape[0]
        nb = L.eonerf_render_workspace_bytes(native, n, flags)
        # a training chunk keeps its own workspace alive until its backward; inference chunks share one
        ws = torch.empty(nb, dtype=torch.uint8, device=table.device) if train else field._workspace("render", nb)
        out = torch.empty(n, 21, dtype=torch.float32, device=table.device)
        n_samples = torch.zeros(1, dtype=torch.int32, device=table.device)
        _lib.check(L.eonerf_render_forward(native, _ptr(flat), _ptr(table), _ptr(img), _ptr(_zsteps(table.device, ns)),
                                           _ptr(u_cam), _ptr(u_retry), _ptr(u_sun), n, flags, _ptr(out), _ptr(n_samples),
                                           _ptr(ws), ws.numel(), _stream()))
        if train:
            ctx.field, ctx.flags, ctx.ws, ctx.ns = field, flags, ws, ns
            ctx.save_for_backward(table, img)
            ctx.mark_non_differentiable(n_samples)
        else:       # inference / only_depth chunks keep nothing for a backward pass: their outputs carry no graph
            ctx.mark_non_differentiable(out, n_samples)
        return out, n_samples

    @staticmethod
    def backward(ctx, d_out, _d_n):
        field, flags, ws = ctx.field, ctx.flags, ctx.ws
        table, img = ctx.saved_tensors
        L = _lib.lib()
        field.set_n_samples(ctx.ns)                             # the backward runs under its forward's sample count
        flat = field.flat_params()
        d_flat = torch.zeros_like(flat)
        d_out = d_out.contiguous().float()
        _lib.check(L.eonerf_render_backward(field._ctx, _ptr(flat), _ptr(table), _ptr(img), table.shape[0], flags,
                                            _ptr(d_out), _ptr(d_flat), _ptr(ws), ws.numel(), _stream()))
        ctx.ws = None
        return (None,) * 8 + tuple(field.grad_views(d_flat))


def render_rays_chunk(radiance_field, table, img, epoch_idx, eval=False, only_depth=False, noise=None):
    """table [n,11] fp32, img [n] int64 -> (out [n,21], n_samples int32[1]) for one chunk."""
    # an EXPORT render: eval=True (eval_eonerf.py:311-324) or a module in .eval() mode outside autograd (train_eonerf.py:197-226)
    export = bool(eval) or (not radiance_field.training and not torch.is_grad_enabled())
    n, dev = table.shape[0], table.device
    flags = 0
    if epoch_idx is not None and epoch_idx >= 2:
        flags |= _lib.F_SHADOWS
    if eval:
        flags |= _lib.F_EVAL
    if only_depth:
        flags |= _lib.F_ONLY_DEPTH
    params = list(radiance_field.parameters())
    if torch.is_grad_enabled() and any(p.requires_grad for p in params) and not only_depth:
        flags |= _lib.F_TRAIN
    if noise is None:       # production: no noise buffers, the sampler kernels draw the jitter (Philox; seed: EONerfMLP.set_noise_seed)
        u_cam = u_retry = u_sun = None
    else:
        u_cam, u_retry, u_sun = (None if t is None else t.to(dev, torch.float32).contiguous() for t in noise)
    return _RenderChunk.apply(radiance_field, table, img, flags, export, u_cam, u_retry, u_sun, *params)


def render_image(
    # scene
    radiance_field,
    occupancy_grid,
    rays: SatRays,
    scene_aabb,
    args,
    epoch_idx=None,
    chunk: int = 5120,
    # rendering options (accepted and ignored exactly as the reference's live sampler ignores them)
    near_plane=None,
    far_plane=None,
    render_step_size: float = 1e-3,
    render_bkgd=None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    early_stop_eps: float = 0.0,
    timestamps=None,
    only_depth: bool = False,
    eval: bool = False,
    noise=None,
):
    """Render the pixels of an image (sat_rendering.py:176-335).  Returns (results dict, n_rendering_samples)."""
    radiance_field._context()
    radiance_field.set_n_samples(n_samples_of(render_step_size))
    rays_shape = rays.origins.shape
    if len(rays_shape) == 3:
        height, width, _ = rays_shape
        num_rays = height * width
        rays = namedtuple_map(lambda r: r.reshape([num_rays] + list(r.shape[2:])), rays)
    else:
        num_rays, _ = rays_shape
    table, img = satrays_to_table(rays)

    export = bool(eval) or (not radiance_field.training and not torch.is_grad_enabled())
    for _attempt in range(2):
        outs, counts = [], []
        for k, i in enumerate(range(0, num_rays, chunk)):
            nz = None if noise is None else noise[k]
            out, n = render_rays_chunk(radiance_field, table[i:i + chunk], img[i:i + chunk], epoch_idx, eval=eval,
                                       only_depth=only_depth, noise=nz)
            outs.append(out)
            counts.append(n)
        out = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        n_rendering_samples = int(torch.stack(counts).sum().item())     # the only host sync of the call
        # export renders on the fp16x3 context: its range check rides on that sync; if an operand left the range the split precision
        # carries, the module has switched to its fp32 export context and the image is rendered once more (EONerfMLP._export_range_ok)
        if not export or radiance_field._export_range_ok():
            break
    if torch.is_grad_enabled() and any(p.requires_grad for p in radiance_field.parameters()):
        # training through autograd: the stream is synchronised right here anyway, so this is where a device-side fault of an EARLIER
        # backward (pipelined kernels' watchdog, include/eonerf_hip.h) surfaces -- before another optimizer step builds on it
        _lib.check(_lib.lib().eonerf_device_status(radiance_field._ctx, _stream()))
    lead = tuple(rays_shape[:-1])
    if only_depth:
        return {"depth": out[:, 3:4].reshape(*lead, -1)}, n_rendering_samples
    results = {k: out[:, a:b].reshape(*lead, -1) for k, a, b in RESULT_SLICES}
    return results, n_rendering_samples
