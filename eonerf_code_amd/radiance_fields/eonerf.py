"""EONerfMLP -- host-side mirror of radiance_fields/eonerf.py:69-248 over libeonerf_hip.so.

Same constructor, attributes, state_dict (44 entries) and method signatures as the reference class; the arithmetic
(encoding, trunk, heads, compositing) runs in hand-written HIP kernels for gfx950.  All parameters are views of ONE
flat fp32 device buffer (`flat_params`), which is what the kernels read, what Adam updates and what the data-parallel
launcher all-reduces.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib
from .mlp import MLP, DenseLayer, SinusoidalEncoder


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _FieldFn(torch.autograd.Function):
    """EONerfMLP.forward / query_density as ONE differentiable op (the reference class is an ordinary autograd module,
    radiance_fields/eonerf.py:141-170): forward = eonerf_field_forward_train (chain kernel in training mode: activations and ReLU
    masks stay in the op's workspace), backward = eonerf_field_backward (backward chain with input gradient, weight-gradient GEMM,
    embedding and ambient-head gradients).  The parameters are inputs so autograd routes their gradients."""

    @staticmethod
    def forward(ctx, field, density_only, x, sun, img, *params):
        L = _lib.lib()
        flat = field._ensure_packed()
        n, dev = x.shape[0], x.device
        nb = L.eonerf_field_train_workspace_bytes(field._ctx, n, 1 if density_only else 0)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)        # lives until backward
        sigma = torch.empty(n, 1, dtype=torch.float32, device=dev)
        if density_only:
            albedo = ambient = ts = tb = None
        else:
            albedo, ambient = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(2))
            ts, tb = (torch.empty(n, 1, dtype=torch.float32, device=dev) for _ in range(2))
        _lib.check(L.eonerf_field_forward_train(field._ctx, _ptr(flat), _ptr(x), _ptr(sun), _ptr(img), n, 1 if density_only else 0,
                                                _ptr(sigma), _ptr(albedo), _ptr(ambient), _ptr(ts), _ptr(tb), _ptr(ws), ws.numel(), _stream()))
        ctx.field, ctx.density_only, ctx.ws, ctx.n = field, density_only, ws, n
        ctx.sun = sun
        ctx.need_dx = x.requires_grad
        return sigma if density_only else (sigma, albedo, ambient, ts, tb)

    @staticmethod
    def backward(ctx, *g):
        field, ws, n = ctx.field, ctx.ws, ctx.n
        if ws is None:
            raise RuntimeError("EONerfMLP: backward through the same forward twice (the op's workspace is released after the first)")
        L = _lib.lib()
        flat = field.flat_params()
        d_flat = torch.zeros_like(flat)
        gs = [None if t is None else t.contiguous().float() for t in g] + [None] * 4
        d_x = torch.empty(n, 3, dtype=torch.float32, device=flat.device) if ctx.need_dx else None
        _lib.check(L.eonerf_field_backward(field._ctx, _ptr(flat), _ptr(ctx.sun), n, 1 if ctx.density_only else 0,
                                           _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), _ptr(gs[4]),
                                           _ptr(d_flat), _ptr(d_x), _ptr(ws), ws.numel(), _stream()))
        ctx.ws = None
        return (None, None, d_x, None, None) + tuple(field.grad_views(d_flat))


class _RenderingFn(torch.autograd.Function):
    """EONerfMLP.rendering / render_depth on caller-provided samples as ONE differentiable op (the reference's methods are ordinary
    autograd code: gather, field forward, nerfacc compositing -- radiance_fields/eonerf.py:172-248): forward = eonerf_rendering_train
    (training-mode chain kernel + per-ray compositing; everything the backward needs stays in the op's workspace), backward =
    eonerf_rendering_backward (compositing backward, heads chain, pipelined trunk, weight-gradient GEMM, embedding / ambient head).
    Gradients flow to the parameters; t_starts / t_ends come from samplers that run under no_grad in the reference and carry none."""

    @staticmethod
    def forward(ctx, field, depth_only, table, img, ts_, te_, ri, n_rays, *params):
        L = _lib.lib()
        flat = field._ensure_packed()
        dev, n = table.device, ts_.shape[0]
        flags = _lib.F_TRAIN | (_lib.F_ONLY_DEPTH if depth_only else 0)
        nb = L.eonerf_render_workspace_bytes(field._ctx, n_rays, flags)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)        # lives until backward
        depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
        if depth_only:
            albedo = beta = tsc = ambient = entropy = None
        else:
            albedo, ambient = (torch.empty(n_rays, 3, dtype=torch.float32, device=dev) for _ in range(2))
            beta, tsc, entropy = (torch.empty(n_rays, 1, dtype=torch.float32, device=dev) for _ in range(3))
        _lib.check(L.eonerf_rendering_train(field._ctx, _ptr(flat), _ptr(table), _ptr(img), _ptr(ts_), _ptr(te_), _ptr(ri), n, n_rays,
                                            1 if depth_only else 0, _ptr(albedo), _ptr(depth), _ptr(beta), _ptr(tsc), _ptr(ambient),
                                            _ptr(entropy), _ptr(ws), ws.numel(), _stream()))
        ctx.field, ctx.depth_only, ctx.ws, ctx.n_rays = field, depth_only, ws, n_rays
        ctx.ns = field._n_samples                                  # the workspace is carved for THIS step size (include/eonerf_hip.h)
        ctx.save_for_backward(table, img)
        if depth_only:
            return depth
        ctx.mark_non_differentiable(entropy)
        return albedo, depth, beta, tsc, ambient, entropy

    @staticmethod
    def backward(ctx, *g):
        field, ws = ctx.field, ctx.ws
        if ws is None:
            raise RuntimeError("EONerfMLP.rendering: backward through the same call twice (the op's workspace is released after the first)")
        table, img = ctx.saved_tensors
        L = _lib.lib()
        field.set_n_samples(ctx.ns)                                # a render at another step size may have run since the forward
        flat = field.flat_params()
        d_flat = torch.zeros_like(flat)
        gs = [None if t is None else t.contiguous().float() for t in g]
        if ctx.depth_only:
            g_alb = g_beta = g_ts = g_amb = None
            g_depth = gs[0]
        else:
            g_alb, g_depth, g_beta, g_ts, g_amb = gs[:5]
        _lib.check(L.eonerf_rendering_backward(field._ctx, _ptr(flat), _ptr(table), _ptr(img), ctx.n_rays, 1 if ctx.depth_only else 0,
                                               _ptr(g_alb), _ptr(g_depth), _ptr(g_beta), _ptr(g_ts), _ptr(g_amb),
                                               _ptr(d_flat), _ptr(ws), ws.numel(), _stream()))
        ctx.ws = None
        return (None,) * 8 + tuple(field.grad_views(d_flat))


class EONerfMLP(nn.Module):
    def __init__(self, n_input_images: int, net_depth: int = 8, net_width: int = 256, skip_layer: int = 4,
                 radiometric_normalization: bool = False, precision: str = None, eval_precision: str = None):
        super().__init__()
        if (net_depth, net_width, skip_layer) != (8, 256, 4):
            raise ValueError("the HIP path implements the shipped EO-NeRF geometry only: depth 8, width 256, skip 4")
        self.pos_enc_L, self.view_enc_L = 10, 4
        self.n_input_images = n_input_images
        self.posi_encoder = SinusoidalEncoder(3, 0, self.pos_enc_L, True)
        self.view_encoder = SinusoidalEncoder(3, 0, self.view_enc_L, True)
        self.transient_encoder = nn.Embedding(n_input_images, 4)
        self.beta_min = 0.05
        self.radiometric_normalization = radiometric_normalization
        if radiometric_normalization:
            init = torch.cat([torch.ones(n_input_images, 3), torch.zeros(n_input_images, 6)], dim=1)
            self.radiometricT_enc = nn.Embedding.from_pretrained(init, freeze=False)
        self.base_mlp = MLP(self.posi_encoder.latent_dim, None, net_depth, net_width, skip_layer, output_enabled=False)
        self.sigma_layer = DenseLayer(self.base_mlp.output_dim, 1)
        self.bottleneck_layer = DenseLayer(self.base_mlp.output_dim, net_width)
        self.albedo_mlp = MLP(net_width, 3, 1, net_width // 2, None)
        self.transient_mlp = MLP(net_width + 4, None, 4, net_width // 2, None, output_enabled=False)
        self.transient_scalar = DenseLayer(self.transient_mlp.output_dim, 1)
        self.transient_beta = DenseLayer(self.transient_mlp.output_dim, 1)
        self.ambient_mlp = MLP(self.view_encoder.latent_dim, 3, 1, net_width // 2, None)

        # "fp16x3": an INFERENCE precision (every operand as hi + lo fp16, three fp16 MFMAs per product: fp32-level accuracy at a fraction
        # of the fp32 path's time); a module created with it renders and evaluates, training calls raise (EONERF_E_UNSUPPORTED)
        self.precision = (precision or os.environ.get("EONERF_PRECISION", "bf16")).lower()
        if self.precision not in _lib.PRECISIONS:
            raise ValueError("precision must be 'bf16', 'fp32' or 'fp16x3' (inference only)")
        # EXPORT renders (render_image(eval=True), or a module in .eval() mode under no_grad: eval_eonerf.py:311-324, the validation
        # of train_eonerf.py:197-226 -- where DSMs and their MAE come from) do not run in bf16 whatever the training precision: on
        # identical weights the fp32-accurate kernels match the reference arithmetic to 1e-4 (altitude well inside 1 cm), while the bf16
        # kernels move the rendered surface of a trained field by ~1 cm per ray (weight rounding of the trunk).
        # "fp16x3" (default since round 4): the split precision above, ~5x faster than "fp32" (exact fp32 FMA chains); "same": no switch.
        self.eval_precision = (eval_precision or os.environ.get("EONERF_EVAL_PRECISION", "fp16x3")).lower()
        if self.eval_precision not in ("fp32", "fp16x3", "same"):
            raise ValueError("eval_precision must be 'fp16x3', 'fp32' or 'same'")
        self._n_samples = 128     # int(2 / render_step_size) of the native contexts' next calls (set_n_samples: 2 .. 256)
        self._ctx = None          # eonerf_ctx*
        self._ctx_eval = None     # second native context (fp32) for export renders of a bf16 field, created on first use
        self._packed_version_eval = None
        self._native_gen = 0      # bumped by whoever changes the parameters through raw pointers (FusedTrainer's Adam kernel: no
        #                           tensor ._version moves), so that every OTHER context over the flat buffer re-packs before its next call
        self._noise_seed = None
        self._flat = None         # flat fp32 parameter buffer (device)
        self._layout = None
        self._packed_version = None
        self._ws = {}

    # ------------------------------------------------------------------ native context / flat parameters
    def _context(self):
        if self._ctx is None:
            L = _lib.lib()
            cfg = _lib.EonerfConfig(self.n_input_images, _lib.PRECISIONS[self.precision], self._n_samples, 1 if self.radiometric_normalization else 0)
            ctx = C.c_void_p()
            _lib.check(L.eonerf_create(C.byref(ctx), C.byref(cfg)))
            self._ctx = ctx
            self._layout = _lib.param_layout(ctx)
            self._n_floats = L.eonerf_param_floats(ctx)
        return self._ctx

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.lib().eonerf_destroy(self._ctx)
            if getattr(self, "_ctx_eval", None) is not None:
                _lib.lib().eonerf_destroy(self._ctx_eval)
        except Exception:
            pass

    def _native(self, export=False):
        """(context, flat parameters) to run a call on, packed weights up to date.  export=True on a bf16 module with
        eval_precision "fp16x3" / "fp32": the module's second context, of that precision, over the SAME flat parameter buffer."""
        if not (export and self.precision == "bf16" and self.eval_precision != "same"):
            return self._context(), self._ensure_packed()
        flat = self.flat_params()
        L = _lib.lib()
        if self._ctx_eval is None:
            cfg = _lib.EonerfConfig(self.n_input_images, _lib.PRECISIONS[self.eval_precision], self._n_samples, 1 if self.radiometric_normalization else 0)
            ctx = C.c_void_p()
            _lib.check(L.eonerf_create(C.byref(ctx), C.byref(cfg)))
            self._ctx_eval = ctx
            if self._noise_seed is not None:
                _lib.check(L.eonerf_set_noise_seed(ctx, self._noise_seed))
        ver = (flat.data_ptr(), self._native_gen) + tuple(p._version for p in self.parameters())
        if ver != self._packed_version_eval:
            _lib.check(L.eonerf_set_weights(self._ctx_eval, _ptr(flat), _stream()))
            self._packed_version_eval = ver
        return self._ctx_eval, flat

    def _export_range_ok(self):
        """Call where an fp16 x 3 call has just been (or may now be) synchronised: False when an operand left the range the split
        precision carries (eonerf_range_status: a weight matrix, an activation or a position; include/eonerf_hip.h) -- its outputs are
        invalid.  A bf16 module then switches its export context to "fp32" for good (the caller repeats the call); a module whose OWN
        precision is fp16x3 has nowhere to go and raises."""
        if self.precision == "fp16x3":
            ctx = self._ctx
        elif self.precision == "bf16" and self.eval_precision == "fp16x3":
            ctx = self._ctx_eval
        else:
            return True
        if ctx is None:
            return True
        L = _lib.lib()
        rc = L.eonerf_range_status(ctx, _stream())
        if rc == 0:
            return True
        if rc != _lib.E_RANGE or self.precision == "fp16x3":
            _lib.check(rc)
        import warnings
        warnings.warn("EONerfMLP: a weight matrix or an activation is outside the range of the fp16x3 export precision "
                      "(|v| > 65504, or a layer with max|w| outside [2^-9, 64]); export renders of this module use the fp32 kernels from now on")
        L.eonerf_destroy(self._ctx_eval)
        self._ctx_eval, self._packed_version_eval = None, None
        self.eval_precision = "fp32"
        return False

    def _named(self):
        return dict(self.named_parameters())

    def flat_params(self):
        """The flat fp32 buffer all parameters are views of (re-established after .to()/load_state_dict rebinding)."""
        self._context()
        params = self._named()
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("EONerfMLP's hot path runs on an AMD GPU only; move the module to cuda first (no CPU fallback)")
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            for name, off, r, c in self._layout:
                if name in params and params[name].data_ptr() != base + 4 * off:
                    ok = False
                    break
        if not ok:
            flat = torch.zeros(self._n_floats, dtype=torch.float32, device=dev)
            for name, off, r, c in self._layout:
                if name not in params:
                    continue
                p = params[name]
                flat[off:off + r * c].copy_(p.data.reshape(-1))
                p.data = flat[off:off + r * c].view(p.shape)
            self._flat = flat
            self._packed_version = None
        return self._flat

    def _ensure_packed(self):
        flat = self.flat_params()
        ver = tuple(p._version for p in self.parameters())
        if ver != self._packed_version:
            _lib.check(_lib.lib().eonerf_set_weights(self._ctx, _ptr(flat), _stream()))
            self._packed_version = ver
        return flat

    def set_n_samples(self, n_samples):
        """n_samples = int(2 / render_step_size) (sat_rendering.py:64, opt.py:54) of the calls that follow: 2 .. 256 -- render_image
        and satnerf_sampling call this with the step size they are given; everything sized 128 / 127 in the library follows it."""
        n_samples = int(n_samples)
        if not 2 <= n_samples <= 256:
            raise ValueError(f"{n_samples} samples per ray: the HIP path supports 2 .. 256 (a ray's samples live in the 64 lanes x 4 slots of one wavefront)")
        if n_samples != self._n_samples:
            self._n_samples = n_samples
            for ctx in (self._ctx, self._ctx_eval):
                if ctx is not None:
                    _lib.check(_lib.lib().eonerf_set_n_samples(ctx, n_samples))

    def set_noise_seed(self, seed):
        """Key of the sampler's in-kernel jitter stream (what torch.manual_seed is to perturb_z_vals' rand_like)."""
        self._context()
        self._noise_seed = int(seed)
        _lib.check(_lib.lib().eonerf_set_noise_seed(self._ctx, self._noise_seed))
        if self._ctx_eval is not None:      # the export context draws from the same stream as the main one
            _lib.check(_lib.lib().eonerf_set_noise_seed(self._ctx_eval, self._noise_seed))

    def weights_changed_natively(self):
        """Tell the module that its flat parameter buffer was updated through raw pointers (a native optimizer step): contexts whose
        packed weight streams were NOT refreshed by that update (the fp32 export context) re-pack before their next call."""
        self._native_gen += 1

    def grad_views(self, d_flat):
        """Views of a flat gradient buffer in named_parameters() order (None for tensors absent from the layout)."""
        by_name = {name: (off, r, c) for name, off, r, c in self._layout}
        out = []
        for name, p in self.named_parameters():
            off, r, c = by_name[name]
            out.append(d_flat[off:off + r * c].view(p.shape))
        return out

    def _workspace(self, key, nbytes):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != self._flat.device:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self._flat.device)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ reference API
    def _wants_grad(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in self.parameters())

    def query_density(self, x):
        """radiance_fields/eonerf.py:141-145: x[..., 3] -> sigma[..., 1].  Differentiable (w.r.t. the parameters and x) when
        autograd is recording; a plain inference call otherwise."""
        shape = x.shape[:-1]
        xs = x.reshape(-1, 3).float().contiguous()
        n = xs.shape[0]
        if n > 0 and self._wants_grad(x):
            self._ensure_packed()
            return _FieldFn.apply(self, True, xs, None, None, *self.parameters()).view(*shape, 1)
        with torch.no_grad():
            for _attempt in range(2):
                # a module in .eval() mode evaluates on the export context, like its renders do (one checkpoint, one arithmetic)
                native, flat = self._native(export=not self.training)
                sigma = torch.empty(n, dtype=torch.float32, device=xs.device)
                L = _lib.lib()
                nb = L.eonerf_field_workspace_bytes(native, n)
                ws = self._workspace("field", nb)
                _lib.check(L.eonerf_query_density(native, _ptr(flat), _ptr(xs), n, _ptr(sigma), _ptr(ws), ws.numel(), _stream()))
                if self.training or self._export_range_ok():      # (fp16x3 export context: range check, once more in fp32 if it fired)
                    break
        return sigma.view(*shape, 1)

    def query_opacity(self, x, step_size):
        """radiance_fields/eonerf.py:147-152."""
        return self.query_density(x) * step_size

    def forward(self, x, sun_dirs=None, img_indices=None):
        """radiance_fields/eonerf.py:154-170: -> (sigma[N,1], albedo[N,3], ambient[N,3], transient_scalar[N,1], transient_beta[N,1]).
        Differentiable (parameters and x) when autograd is recording, as the reference module is; the training loop itself
        differentiates through render_image (sat_rendering.py), which fuses sampling, both passes and compositing."""
        xs = x.reshape(-1, 3).float().contiguous()
        n = xs.shape[0]
        sun = sun_dirs.reshape(-1, 3).float().contiguous()
        img = img_indices.reshape(-1).to(torch.int64).contiguous()
        if n > 0 and self._wants_grad(x):
            self._ensure_packed()
            return _FieldFn.apply(self, False, xs, sun, img, *self.parameters())
        with torch.no_grad():
            for _attempt in range(2):
                native, flat = self._native(export=not self.training)
                dev = xs.device
                sigma, ts, tb = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
                albedo, ambient = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(2))
                L = _lib.lib()
                nb = L.eonerf_field_workspace_bytes(native, n)
                ws = self._workspace("field", nb)
                _lib.check(L.eonerf_field_forward(native, _ptr(flat), _ptr(xs), _ptr(sun), _ptr(img), n, _ptr(sigma), _ptr(albedo),
                                                  _ptr(ambient), _ptr(ts), _ptr(tb), _ptr(ws), ws.numel(), _stream()))
                if self.training or self._export_range_ok():
                    break
        return sigma.view(n, 1), albedo, ambient, ts.view(n, 1), tb.view(n, 1)

    def _rendering(self, chunk_rays, t_starts, t_ends, ray_indices, depth_only):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and t_starts.shape[0] > 0:
            # the reference's rendering() / render_depth() build an autograd graph (radiance_fields/eonerf.py:172-248): one differentiable op
            from ..datasets.satellite import satrays_to_table
            self._ensure_packed()
            table, img = satrays_to_table(chunk_rays)
            ts_, te_ = t_starts.detach().float().contiguous(), t_ends.detach().float().contiguous()
            ri = ray_indices.to(torch.int64).contiguous()
            out = _RenderingFn.apply(self, depth_only, table, img, ts_, te_, ri, table.shape[0], *self.parameters())
            self._patch_t_ends(t_ends, ri)
            return (None, out, None, None, None, None) if depth_only else out
        with torch.no_grad():
            return self._rendering_impl(chunk_rays, t_starts, t_ends, ray_indices, depth_only)

    @staticmethod
    def _patch_t_ends(t_ends, ri):
        """the reference patches the caller's t_ends in place (eonerf.py:218-220): last interval of every ray -> 1e10"""
        n = t_ends.shape[0]
        if n > 0:
            with torch.no_grad():
                last = torch.ones(n, dtype=torch.bool, device=t_ends.device)
                last[:-1] = ri[1:] != ri[:-1]
                t_ends[last] = 1e10

    def _rendering_impl(self, chunk_rays, t_starts, t_ends, ray_indices, depth_only):
        from ..datasets.satellite import satrays_to_table
        flat = self._ensure_packed()
        table, img = satrays_to_table(chunk_rays)
        n_rays, n = table.shape[0], t_starts.shape[0]
        dev = table.device
        ts_, te_ = t_starts.float().contiguous(), t_ends.float().contiguous()
        ri = ray_indices.to(torch.int64).contiguous()
        depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
        if depth_only:
            albedo = beta = tsc = ambient = entropy = None
        else:
            albedo, ambient = (torch.empty(n_rays, 3, dtype=torch.float32, device=dev) for _ in range(2))
            beta, tsc, entropy = (torch.empty(n_rays, 1, dtype=torch.float32, device=dev) for _ in range(3))
        L = _lib.lib()
        nb = L.eonerf_render_workspace_bytes(self._ctx, n_rays, _lib.F_ONLY_DEPTH if depth_only else 0)
        ws = self._workspace("render", nb)
        _lib.check(L.eonerf_rendering(self._ctx, _ptr(flat), _ptr(table), _ptr(img), _ptr(ts_), _ptr(te_), _ptr(ri), n, n_rays,
                                      1 if depth_only else 0, _ptr(albedo), _ptr(depth), _ptr(beta), _ptr(tsc), _ptr(ambient),
                                      _ptr(entropy), _ptr(ws), ws.numel(), _stream()))
        self._patch_t_ends(t_ends, ri)
        return albedo, depth, beta, tsc, ambient, entropy

    def render_depth(self, chunk_rays, t_starts, t_ends, ray_indices):
        """radiance_fields/eonerf.py:172-194: flattened samples -> depth [n_rays, 1] (differentiable w.r.t. the parameters)."""
        return self._rendering(chunk_rays, t_starts, t_ends, ray_indices, True)[1]

    def rendering(self, chunk_rays, t_starts, t_ends, ray_indices, epoch_idx=100):
        """radiance_fields/eonerf.py:196-248: -> (albedo_rgb_, depth_, transient_beta_, transient_scalar_, ambient_rgb_, entropy_).
        Differentiable w.r.t. the parameters when autograd is recording, as the reference method is (round 4); render_image fuses this
        with sampling, the shadow pass and autograd and is what the training loop calls.  At most 127 samples per ray, ray_indices
        sorted (what satnerf_sampling produces)."""
        return self._rendering(chunk_rays, t_starts, t_ends, ray_indices, False)
