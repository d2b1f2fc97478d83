"""Fused training step over libeonerf_hip.so -- the loop body of train_eonerf.py:104-161 without autograd plumbing:

    jitter noise -> eonerf_render_forward(TRAIN) -> loss gradient on [R,21] -> eonerf_render_backward -> eonerf_grad_seal
      -> [RCCL all-reduce of the flat gradient message] -> eonerf_adam_step (+ weight re-pack)

One process per GPU; with torch.distributed initialised (backend "nccl" = RCCL over xGMI) the flat fp32 gradient
(679,821 floats at 20 images, 2.7 MB) is summed across ranks in ONE collective per step and scaled by 1/world inside
the Adam kernel.  Rays are independent units, so ranks render disjoint ray batches and nothing else is exchanged.

Fault protocol (device-side watchdog of the pipelined backward, include/eonerf_hip.h): the message carries one control float
behind the gradients -- "my gradients are invalid" -- written by eonerf_grad_seal from the context's sticky status word.  The
sum all-reduce hands every rank the OR of all flags, the Adam kernel skips the update on every rank while it (or the local status
word) is set, and every rank raises at its next check_device_status(): replicas never apply a poisoned gradient and never diverge.
"""
import ctypes as C
import os

import torch

from . import _lib
from .radiance_fields.eonerf import EONerfMLP, _ptr, _stream
from .sat_rendering import _zsteps


def reduce_gradients(d_flat):
    """The ONE exchange step of data-parallel training: sum the flat fp32 gradient over all ranks (RCCL on GPUs, gloo in
    the CPU tests).  Returns the factor the optimizer must scale it by (1/world: every rank holds the mean-loss gradient
    of its own equally-sized batch, so the mean over ranks is the gradient of the global-batch mean loss)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    if dist.get_world_size() == 1 and os.environ.get("EONERF_FORCE_ALLREDUCE") != "1":     # (test hook: run the collective at world 1 too)
        return 1.0
    dist.all_reduce(d_flat, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def rank_slice(n_total, rank, world):
    """Contiguous share of a shared permutation that rank `rank` consumes (equal sizes; the tail is dropped as
    DataLoader(drop_last) would)."""
    per = n_total // world
    return rank * per, (rank + 1) * per


class FusedTrainer:
    def __init__(self, field: EONerfMLP, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, max_rays: int = 4096,
                 keep_message: bool = True, n_samples: int = 128):
        """keep_message=True: the gradient message survives the update (d_flat can be read after step()) and is sealed with the
        fault flag even when there are no peers -- what tests and debugging want.  The launcher and bench.py pass False: the update
        consumes the message (eonerf_adam_step_zero_grad: optimizer.step() + the next optimizer.zero_grad(), train_eonerf.py:158-161,
        in one kernel) and a single process skips the seal (k_adam reads the status word itself): two launches fewer per step."""
        self.field = field
        self.keep_message = keep_message
        self.tail_events = None      # a list: reduce_and_update() appends an event pair around the exchange + update (bench.py, N > 1)
        self.presample_events = []   # ... and one around the next step's sampler when it was enqueued under that exchange
        self._hint = None            # (rays, img_idx, their version counters) of the batch eonerf_presample last ran on
        # EONERF_EXCHANGE_BUCKETS (default 2; 1 = the single all-reduce behind the last gradient kernel): with 2 the EARLY block of the
        # message -- the trunk layers the camera pass' pipelined launch completes, 58 % of it (eonerf_grad_early_floats) -- is all-reduced
        # on the communication stream as soon as that launch has ended, under the weight-gradient GEMM and the tail (~0.5 ms); the rest
        # (+ the fault flag) follows behind the last gradient kernel; Adam waits for both.  EONERF_COMM_CUS (default 8): CUs the gradient
        # kernels behind that point leave free for the collective's kernel (they fill every CU they are given)
        self.buckets = 2 if os.environ.get("EONERF_EXCHANGE_BUCKETS", "2") != "1" else 1
        self.comm_cus = int(os.environ.get("EONERF_COMM_CUS", "8"))
        self.hidden_events = None    # a list: (start of the early collective, its end, end of the backward) per step (bench.py, N > 1)
        self._early_event = None
        self.exchange_async = os.environ.get("EONERF_EXCHANGE", "side") == "async"      # A/B: all_reduce(async_op=True) instead of a side stream
        self.presample = os.environ.get("EONERF_PRESAMPLE", "1") != "0"          # next step's sampler under the gradient exchange (N > 1)
        self.fused_loss = os.environ.get("EONERF_FUSED_LOSS", "1") != "0"      # (A/B and test switch: 0 = eonerf_train_loss + eonerf_render_backward)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.flat = field._ensure_packed()
        self.n_samples_per_ray = int(n_samples)      # int(2 / render_step_size), train_eonerf.py:50-53: 2 .. 256
        field.set_n_samples(self.n_samples_per_ray)
        dev = self.flat.device
        self.L = _lib.lib()
        self.ctx = field._ctx
        self.n_params = self.flat.numel()
        # the gradient MESSAGE: parameters' gradients + 4 control floats ([n_params] = fault flag), one all-reduce unit
        self.d_flat = torch.zeros(self.L.eonerf_grad_floats(self.ctx), dtype=torch.float32, device=dev)
        self._grad_clean = True      # d_flat holds zeros (fresh, or consumed by the last update)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0     # ONE Adam step count for every parameter (zero gradients still step, see k_adam)
        self.dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.dist_on else 1
        self.max_rays = max_rays
        self._ws = {}
        self._comm_stream = None
        self.out = torch.empty(max_rays, 21, dtype=torch.float32, device=dev)
        self.d_out = torch.zeros(max_rays, 21, dtype=torch.float32, device=dev)
        self.n_samples = torch.zeros(1, dtype=torch.int32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.zsteps = _zsteps(dev, self.n_samples_per_ray)
        self.n_early = int(self.L.eonerf_grad_early_floats(self.ctx))
        self.set_exchange_buckets(self.buckets)
        if self.world > 1:     # identical replicas: broadcast rank 0's parameters once (train_eonerf.py has one process)
            torch.distributed.broadcast(self.flat, src=0)
            _lib.check(self.L.eonerf_set_weights(self.ctx, _ptr(self.flat), _stream()))

    def set_lr(self, lr):
        """StepLR(gamma=0.9) is applied by the caller once per epoch (train_eonerf.py:64,304)."""
        self.lr = lr

    def _workspace(self, n, flags):
        key = (n, flags)
        ws = self._ws.get(key)
        if ws is None:
            nb = self.L.eonerf_render_workspace_bytes(self.ctx, n, flags)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.flat.device)
            self._ws = {key: ws}
        return ws

    def loss_grad(self, out, pixels, epoch_idx, d_out):
        """train_eonerf.py:139-143: MSE for epoch < 2, metrics.uncertainty_aware_loss (metrics.py:17-22) afterwards.
        One kernel writes dL/d out into d_out (only the rgb and beta columns are non-zero) and the loss scalar."""
        n = out.shape[0]
        kind = 0 if epoch_idx < 2 else 1
        _lib.check(self.L.eonerf_train_loss(self.ctx, _ptr(out), _ptr(pixels), n, kind, _ptr(d_out), _ptr(self.loss), _stream()))
        return self.loss

    def step(self, rays, img_idx, pixels, epoch_idx, noise=None, profile=False, next_batch=None, aux_loss=None):
        """rays [n,11] fp32, img_idx [n] int64, pixels [n,3] (all on the GPU).  Returns the loss as a device scalar.
        next_batch = (rays, img_idx, epoch_idx[, with_aux_loss]) of the FOLLOWING step, if the caller knows it (RayTable does): at N > 1
        its camera sampler runs under this step's gradient exchange (reduce_and_update).
        aux_loss: see forward_backward."""
        loss = self.forward_backward(rays, img_idx, pixels, epoch_idx, noise, aux_loss)
        self.reduce_and_update(next_batch)
        return loss

    def forward_backward(self, rays, img_idx, pixels, epoch_idx, noise=None, aux_loss=None):
        """First half of a step: render, loss, backward into the gradient message (sealed with this rank's fault flag).
        aux_loss: optional callable `out[n,21] -> scalar` in plain PyTorch for the reference's auxiliary terms (train_eonerf.py:145-155:
        metrics.depth_loss_L2 on out[:, 3], metrics.shadow_loss_L2 on out[:, 10], already gated by the caller's epoch window as
        update_loss_with_aux_term does).  Its gradient w.r.t. the packed outputs (torch autograd over O(n_rays) element-wise work) is
        added to the main loss' gradient before the HIP backward; the returned loss is the sum.  With an auxiliary term the epoch < 2
        graph pruning (EONERF_F_RGB_LOSS: "the loss depends on rgb / depth / albedo only") is not applied: the term may read any column."""
        n = rays.shape[0]
        if n > self.max_rays:
            raise ValueError(f"batch of {n} rays exceeds the trainer's max_rays={self.max_rays}")
        if not (rays.is_cuda and rays.dtype == torch.float32 and rays.dim() == 2 and rays.shape[1] == 11 and rays.is_contiguous()):
            raise ValueError("rays must be a contiguous CUDA fp32 [n, 11] table")
        if not (img_idx.is_cuda and img_idx.dtype == torch.int64 and img_idx.is_contiguous() and img_idx.numel() == n):
            raise ValueError("img_idx must be a contiguous CUDA int64 tensor with one entry per ray")
        if not (pixels.is_cuda and pixels.dtype == torch.float32 and pixels.shape == (n, 3)):
            raise ValueError("pixels must be a CUDA fp32 [n, 3] tensor")
        # the packed weight streams follow every in-place change of the parameters (load_state_dict, load_checkpoint, manual
        # init) and a re-bound flat buffer (.to()); adam_step re-packs them itself
        flat = self.field._ensure_packed()
        if flat.data_ptr() != self.flat.data_ptr():
            if flat.numel() != self.flat.numel() or flat.device != self.flat.device:
                raise RuntimeError("the field's parameters moved to another device after the trainer was built")
            self.flat = flat
        # epoch < 2: s = 1 and the loss is MSE on rgb, so the transient head is outside the autograd graph (F_RGB_LOSS)
        flags = _lib.F_TRAIN | (_lib.F_SHADOWS if epoch_idx >= 2 else (_lib.F_RGB_LOSS if aux_loss is None else 0))
        if noise is None:       # production: the sampler kernels draw the jitter themselves (Philox)
            u_cam = u_retry = u_sun = None
        else:
            u_cam, u_retry, u_sun = noise
        self.field.set_n_samples(self.n_samples_per_ray)      # (another caller of the module may have rendered at another step size)
        hint, self._hint = self._hint, None
        if hint is not None and (hint[0].data_ptr() != rays.data_ptr() or hint[1].data_ptr() != img_idx.data_ptr() or hint[0].shape != rays.shape
                                 or hint[2] != rays._version or hint[3] != img_idx._version):
            # the presampled record is not for THESE tensors as they are now (other batch, or the same buffers refilled in place): drop it,
            # the forward samples itself.  (What torch cannot see -- a raw-pointer write -- the library's device-side digest catches.)
            _lib.check(self.L.eonerf_presample_cancel(self.ctx))
        self._render_forward(rays, img_idx, n, flags, (u_cam, u_retry, u_sun))
        if aux_loss is not None:
            loss = self.loss_grad(self.out[:n], pixels.contiguous(), epoch_idx, self.d_out)
            o = self.out[:n].detach().clone().requires_grad_(True)
            with torch.enable_grad():
                extra = aux_loss(o)
                (g,) = torch.autograd.grad(extra, o)
            self.d_out[:n] += g
            self._render_backward(rays, img_idx, n, flags)
            self.loss_total = loss + extra.detach()
            return self.loss_total
        if self.fused_loss:      # loss + backward in one library call: d out[R,21] is never written (eonerf_render_backward_loss)
            self._render_backward(rays, img_idx, n, flags, pixels=pixels.contiguous(), kind=0 if epoch_idx < 2 else 1)
            return self.loss
        loss = self.loss_grad(self.out[:n], pixels.contiguous(), epoch_idx, self.d_out)
        self._render_backward(rays, img_idx, n, flags)
        return loss

    def _render_forward(self, rays, img_idx, n, flags, noise):
        ws, st = self._workspace(n, flags), _stream()
        _lib.check(self.L.eonerf_render_forward(self.ctx, _ptr(self.flat), _ptr(rays), _ptr(img_idx), _ptr(self.zsteps),
                                                _ptr(noise[0]), _ptr(noise[1]), _ptr(noise[2]), n, flags, _ptr(self.out), _ptr(self.n_samples),
                                                _ptr(ws), ws.numel(), st))

    def _render_backward(self, rays, img_idx, n, flags, pixels=None, kind=0):
        ws, st = self._workspace(n, flags), _stream()
        if not self._grad_clean:      # (the update consumes the message: eonerf_adam_step_zero_grad)
            self.d_flat.zero_()
        self._grad_clean = False
        if pixels is not None:
            _lib.check(self.L.eonerf_render_backward_loss(self.ctx, _ptr(self.flat), _ptr(rays), _ptr(img_idx), n, flags, _ptr(self.out), _ptr(pixels),
                                                          kind, _ptr(self.d_out), _ptr(self.loss), _ptr(self.d_flat), _ptr(ws), ws.numel(), st))
        else:
            _lib.check(self.L.eonerf_render_backward(self.ctx, _ptr(self.flat), _ptr(rays), _ptr(img_idx), n, flags, _ptr(self.d_out),
                                                     _ptr(self.d_flat), _ptr(ws), ws.numel(), st))
        if self.keep_message or self._exchanges():     # the flag travels with the message; alone, k_adam reads the status word itself
            _lib.check(self.L.eonerf_grad_seal(self.ctx, _ptr(self.d_flat), st))

    def set_exchange_buckets(self, n):
        """2: the early block of the gradient message (eonerf_grad_early_floats) is all-reduced from the event the library records behind the
        camera pass' pipelined backward, the rest behind the backward's last kernel; 1: one all-reduce there.  Takes effect with the next
        step; only a trainer that exchanges on its side stream has two buckets (EONERF_EXCHANGE_BUCKETS sets the default)."""
        self.buckets = 2 if n == 2 else 1
        if self.buckets == 2 and self._exchanges() and not self.exchange_async:
            if self._early_event is None:
                self._early_event = torch.cuda.Event()
                self._early_event.record()          # (the HIP event exists from its first record on)
            _lib.check(self.L.eonerf_set_exchange_event(self.ctx, C.c_void_p(self._early_event.cuda_event), self.comm_cus))
        else:
            _lib.check(self.L.eonerf_set_exchange_event(self.ctx, None, 0))
            self._early_event = None

    def _exchanges(self):
        return self.dist_on and (self.world > 1 or os.environ.get("EONERF_FORCE_ALLREDUCE") == "1")

    def reduce_and_update(self, next_batch=None):
        """Second half of a step: the one exchange (sum all-reduce of the message, side stream) and the fused Adam update, which
        the device skips on every rank when any rank sealed a fault into the message.  With next_batch = (rays, img_idx, epoch_idx)
        the next step's weight-independent kernels (camera sampler: count + emit) are enqueued on the compute stream BEHIND the start
        of the exchange and IN FRONT of the update that waits for it (eonerf_presample; EONERF_PRESAMPLE=0 switches it off); the
        forward of that batch must be the trainer's next render, with the very same tensors."""
        st = _stream()
        tail = self.tail_events
        if tail is not None:      # bench.py at N > 1: the serial tail of a step (exchange + update + re-pack) between two events
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        gscale = self._reduce(st, next_batch)
        self.step_count += 1
        flag = C.c_void_p(self.d_flat.data_ptr() + 4 * self.n_params) if (self.keep_message or self._exchanges()) else None
        adam = self.L.eonerf_adam_step if self.keep_message else self.L.eonerf_adam_step_zero_grad
        _lib.check(adam(self.ctx, _ptr(self.flat), _ptr(self.d_flat), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                        self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, gscale, flag, st))
        self._grad_clean = not self.keep_message
        self.field._packed_version = tuple(p._version for p in self.field.parameters())   # adam_step re-packed the weights of self.ctx
        self.field.weights_changed_natively()      # ... and of no other context: the fp32 export context re-packs at its next render
        if tail is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tail.append((e0, e1))

    def _presample(self, rays, img_idx, epoch_idx, with_aux_loss=False):
        n = rays.shape[0]
        if n > self.max_rays or not (rays.is_cuda and rays.dtype == torch.float32 and rays.dim() == 2 and rays.shape[1] == 11 and rays.is_contiguous()
                                     and img_idx.is_cuda and img_idx.dtype == torch.int64 and img_idx.is_contiguous() and img_idx.numel() == n):
            return          # (forward_backward raises for these; here the batch is only a hint)
        flags = _lib.F_TRAIN | (_lib.F_SHADOWS if epoch_idx >= 2 else (0 if with_aux_loss else _lib.F_RGB_LOSS))      # as forward_backward
        self.field.set_n_samples(self.n_samples_per_ray)
        ws = self._workspace(n, flags)
        _lib.check(self.L.eonerf_presample(self.ctx, _ptr(rays), _ptr(img_idx), _ptr(self.zsteps), n, flags, _ptr(self.n_samples),
                                           _ptr(ws), ws.numel(), _stream()))
        # the library matches the record to its forward by POINTER; the contents are ours to vouch for: remember the tensors' version
        # counters (an in-place refill of a reused staging tensor bumps them; views of one table share their base's counter, so another
        # slice object over the same rows compares equal) and keep the tensors alive (a freed batch re-allocated at the same address would
        # pass the pointer test)
        self._hint = (rays, img_idx, rays._version, img_idx._version)

    def _reduce(self, st, next_batch=None):
        """The gradient all-reduce on a SIDE stream: it starts when the last gradient kernel of the backward has
        finished (event on the compute stream) and the Adam kernel waits for it; host-side launches of the next kernels are not
        held up by the collective.  Single process: no-op."""
        if not self.dist_on or (self.world == 1 and os.environ.get("EONERF_FORCE_ALLREDUCE") != "1"):
            return 1.0
        hint = next_batch is not None and self.presample
        if self.exchange_async:
            # the process group's own collective stream IS the side stream: async_op makes it wait for the compute stream as of NOW (the
            # message is final), wait() makes the compute stream wait for the collective -- no stream of ours, two cross-stream hops fewer
            work = torch.distributed.all_reduce(self.d_flat, op=torch.distributed.ReduceOp.SUM, async_op=True)
            if hint:
                self._presample_timed(next_batch)         # compute stream, behind the point the collective waits for: runs beside it
            work.wait()
            return 1.0 / self.world
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.flat.device)
        cur = torch.cuda.current_stream()
        if self._early_event is not None:
            # first bucket: the library recorded _early_event inside the backward, where [0, n_early) became final.  The host enqueues
            # this behind the whole backward, the GPU runs it as soon as the event fires (the queue is a step ahead of the device)
            self._comm_stream.wait_event(self._early_event)
            ev = self.hidden_events
            with torch.cuda.stream(self._comm_stream):
                if ev is not None:
                    a0 = torch.cuda.Event(enable_timing=True); a0.record()
                torch.distributed.all_reduce(self.d_flat[:self.n_early], op=torch.distributed.ReduceOp.SUM)
                if ev is not None:
                    a1 = torch.cuda.Event(enable_timing=True); a1.record()
            if ev is not None:
                b = torch.cuda.Event(enable_timing=True); b.record()          # compute stream: the end of the backward (+ seal)
                ev.append((a0, a1, b))
        self._comm_stream.wait_stream(cur)                   # (an event on the compute stream: the message is final here)
        if hint:
            # the next step's camera sampler, on the compute stream BEHIND that event: it runs while the collective does.  Enqueued
            # first because a host-blocking backend (gloo rehearsals) would otherwise hold it back until the exchange is over
            self._presample_timed(next_batch)
        with torch.cuda.stream(self._comm_stream):
            gscale = reduce_gradients(self.d_flat if self._early_event is None else self.d_flat[self.n_early:])
        cur.wait_stream(self._comm_stream)
        return gscale

    def _presample_timed(self, next_batch):
        ev = self.tail_events is not None
        if ev:
            p0 = torch.cuda.Event(enable_timing=True)
            p0.record()
        self._presample(*next_batch)
        if ev:          # bench.py: how much of the measured tail is next-step work that ran under the exchange
            p1 = torch.cuda.Event(enable_timing=True)
            p1.record()
            self.presample_events.append((p0, p1))

    def check_device_status(self):
        """Synchronises and raises (on THIS rank; call it on every rank) if a device-side hand-off of the pipelined backward timed out
        on any rank since the last check (eonerf_device_status: the status word is sticky and the Adam kernel has been skipping the
        update since); meant for the places where the host reads the loss anyway."""
        _lib.check(self.L.eonerf_device_status(self.ctx, _stream()))

    def set_noise_seed(self, seed):
        """Key of the in-kernel jitter stream; data-parallel ranks must use different seeds (train_dp.py: seed + rank)."""
        self.field.set_noise_seed(seed)

    # ---- measurement hooks ----
    def clock_probe(self):
        """One launch of the library's fixed MFMA loop (eonerf_clock_probe), read back: {"mhz": shader clock the chip held over the probe,
        "us": the probe's duration}.  Synchronises; bench.py calls it outside its timed bracket."""
        if not hasattr(self, "_probe"):
            self._probe = torch.zeros(4, dtype=torch.float32, device=self.flat.device)
        _lib.check(self.L.eonerf_clock_probe(self.ctx, _ptr(self._probe), _stream()))
        cyc, ticks, mhz, _ = self._probe.tolist()
        return {"mhz": mhz, "us": ticks * 0.01}

    def profile_enable(self, max_launches):
        _lib.check(self.L.eonerf_profile_enable(self.ctx, max_launches))

    def profile_read(self):
        """{scope name: (summed ms, launches)} of the library's per-kernel event scopes (include/eonerf_hip.h)."""
        res, k = {}, 0
        while True:
            name = self.L.eonerf_profile_name(k)
            if name is None:
                return res
            ms, cnt = C.c_float(), C.c_int()
            _lib.check(self.L.eonerf_profile_read(self.ctx, k, C.byref(ms), C.byref(cnt)))
            res[name.decode()] = (ms.value, cnt.value)
            k += 1


class RayTable:
    """GPU-resident ray table + on-device batch gather, replacing the reference's per-item DataLoader
    (datasets/satellite.py:806-811, train_eonerf.py:70,99-109: ~1000 __getitem__ calls and one H2D copy per step).
    Every rank holds the full table and walks its own slice of ONE shared per-epoch permutation (same seed everywhere) --
    what DataLoader(shuffle=True) does on a single GPU, sharded like a DistributedSampler."""

    def __init__(self, rays, img_idx, rgbs, device, seed=42, rank=0, world=1):
        self.rays = rays.to(device, torch.float32).contiguous()
        self.img = img_idx.reshape(-1).to(device, torch.int64).contiguous()
        self.rgbs = rgbs.to(device, torch.float32).contiguous()
        self.n, self.seed, self.rank, self.world = self.rays.shape[0], seed, rank, world
        self._perm_epoch, self._perm, self._shuffled = None, None, None

    def steps_per_epoch(self, batch_per_rank):
        return self.n // (batch_per_rank * self.world)

    def batch(self, epoch, step, batch_per_rank):
        """Batch `step` of epoch `epoch` for this rank.  The shuffle is ONE on-device gather of the whole table per epoch (three
        index_select launches over the table, microseconds at 16 MB); a step's batch is then a contiguous slice of the shuffled
        copy -- no per-step gather kernels, no host work beyond slicing.  The shuffled copy doubles the table's footprint
        (60 B per ray: 1.2 GB -> 2.4 GB for 20 M rays, of 288 GB)."""
        if self._perm_epoch != epoch:
            # drawn ON the device (same seed -> same permutation on every rank: one device type, one torch build): no host-side
            # randperm, no H2D copy, no synchronisation at the epoch boundary
            g = torch.Generator(device=self.rays.device).manual_seed(self.seed + epoch)
            self._perm = torch.randperm(self.n, generator=g, device=self.rays.device)
            self._shuffled = (self.rays.index_select(0, self._perm), self.img.index_select(0, self._perm), self.rgbs.index_select(0, self._perm))
            self._perm_epoch = epoch
        lo = (step * self.world + self.rank) * batch_per_rank
        r, i, c = self._shuffled
        return r[lo:lo + batch_per_rank], i[lo:lo + batch_per_rank], c[lo:lo + batch_per_rank]
