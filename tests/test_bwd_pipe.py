"""GPU: the layer-pipelined trunk backward (csrc/eonerf_bwd_pipe.hip, bf16 camera pass) against the chain + GEMM path it replaces.

Both paths run the SAME bf16 arithmetic for the dX chain (same MFMA accumulation order, same rounding points, same ReLU masks), so
dY_l is bit-identical and the parameter gradients differ only by the order of the fp32 sums over samples: <= 1e-4 relative L2 per
tensor (measured ~1e-6).  The watchdog test stalls one stage on purpose: the launch must drain and the failure must be reported."""
import os

import pytest
import torch

from oracle import eonerf_oracle as orc

pytestmark = pytest.mark.gpu
N_IMG = 19


def _field(pipe, seed=7, fault=None, xcd=False):
    """pipe: the layer-pipelined trunk backward (the heads stay in the heads chain + GEMM jobs); xcd: its XCD-local layout (EONERF_PIPE_XCD)."""
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    sd = orc.random_state_dict(N_IMG, seed=seed, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16")
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    old = {k: os.environ.get(k) for k in ("EONERF_PIPE", "EONERF_PIPE_FAULT", "EONERF_PIPE_XCD")}
    os.environ["EONERF_PIPE"] = "1" if pipe else "0"
    os.environ["EONERF_PIPE_XCD"] = "1" if xcd else "0"
    if fault is not None:
        os.environ["EONERF_PIPE_FAULT"] = str(fault)
    try:
        f._context()                      # the library reads the switches when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return f


def _grads(f, R, epoch, seed=3):
    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.trainer import FusedTrainer
    rays, img, rgbs = (t.cuda() for t in synthetic_batch(R, N_IMG, seed=seed))
    g = torch.Generator(device="cuda").manual_seed(seed)
    noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
    tr = FusedTrainer(f, lr=0.0, max_rays=R)
    loss = float(tr.step(rays, img, rgbs, epoch, noise=noise))
    tr.check_device_status()
    return loss, tr.d_flat.clone(), tr


@pytest.mark.parametrize("R,epoch", [(4096, 0), (4096, 3), (300, 0), (37, 3), (1, 0)])
def test_pipelined_backward_matches_chain_plus_gemm(R, epoch):
    """Same bf16 arithmetic on both sides (same rounding points: every dY is rounded to bf16 where it is handed on, every product
    accumulates in fp32); what differs is the summation order of the weight gradients: 1e-4 per tensor."""
    f_old, f_new = _field(False), _field(True)
    l0, g0, _ = _grads(f_old, R, epoch)
    l1, g1, _ = _grads(f_new, R, epoch)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)             # the loss itself is an atomic sum over rays
    assert torch.isfinite(g1).all()
    for (name, p), a, b in zip(f_old.named_parameters(), f_old.grad_views(g0), f_new.grad_views(g1)):
        assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10, (name, (a - b).norm().item(), a.norm().item())


@pytest.mark.parametrize("R,epoch", [(4096, 3), (300, 0)])
def test_xcd_local_pipelines_give_the_same_gradients(R, epoch):
    """EONERF_PIPE_XCD=1 (round 6, off by default: measured neutral, profiles/r06_xcd_local_pipelines.txt): roles by HW_REG_XCC_ID behind a
    rendezvous of the grid, pipelines formed inside an XCD hand their tiles over with default-policy stores (they stay in the XCD's L2),
    the left-over workgroups form cross-XCD pipelines with the write-through protocol.  Placement and hand-off policy only: the same
    gradients as the default layout (summation order of the per-pipeline partial sums differs: 1e-4 per tensor)."""
    f_a, f_b = _field(True), _field(True, xcd=True)
    l0, g0, _ = _grads(f_a, R, epoch)
    l1, g1, _ = _grads(f_b, R, epoch)
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and torch.isfinite(g1).all()
    for (name, p), a, b in zip(f_a.named_parameters(), f_a.grad_views(g0), f_b.grad_views(g1)):
        assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10, (name, (a - b).norm().item(), a.norm().item())


def test_pipelined_backward_is_repeatable_over_many_steps():
    # uneven load, warm caches, ring slots reused hundreds of times: every step must reproduce the first one's gradient
    f = _field(True)
    _, g_ref, tr = _grads(f, 4096, 0)
    from eonerf_code_amd.synthetic import synthetic_batch
    rays, img, rgbs = (t.cuda() for t in synthetic_batch(4096, N_IMG, seed=3))
    g = torch.Generator(device="cuda").manual_seed(3)
    noise = tuple(torch.rand(4096, 128, device="cuda", generator=g) for _ in range(3))
    for it in range(40):
        tr.step(rays, img, rgbs, 0, noise=noise)
        if it % 8 == 7:
            tr.check_device_status()
            assert (tr.d_flat - g_ref).norm().item() <= 2e-6 * g_ref.norm().item(), it


def test_watchdog_drains_the_launch_quickly_reports_and_gates_the_update():
    import time
    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.trainer import FusedTrainer
    f = _field(True, fault=3)               # stage 3 (layer 4) never publishes its tiles
    R = 4096
    rays, img, rgbs = (t.cuda() for t in synthetic_batch(R, N_IMG, seed=3))
    tr = FusedTrainer(f, lr=1e-3, max_rays=R)
    p0, m0 = tr.flat.detach().clone(), tr.exp_avg.clone()
    tr.step(rays, img, rgbs, 0)                           # (first call: lazy allocations, code-object load)
    torch.cuda.synchronize()
    t0 = time.time()
    tr.step(rays, img, rgbs, 0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    # at most ONE 0.3 s timeout (the stage behind the stalled edge; none at all once the sticky status word is up: a slow path that sees it
    # leaves at once), then every other waiting stage leaves the same way: the launch drains in well under a second, not in one timeout
    # per wait and stage (round 2: ~0.3 s x 2 waits x 7 steps per stage, cascading over the edges)
    assert dt < 1.5, dt
    # the corrupted step was NOT applied: the Adam kernel saw the status word
    assert torch.equal(tr.flat.detach(), p0) and torch.equal(tr.exp_avg, m0)
    assert tr.d_flat[tr.n_params].item() == 1.0          # and the gradient message carries the fault flag for the other ranks
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        tr.check_device_status()
    # the context has left the pipelined path with the reported fault (eonerf_device_status): the next step runs through the chain + GEMM
    # backward -- no timeout, clean status, update applied -- although the stalled stage is still armed
    tr.profile_enable(4)
    tr.step(rays, img, rgbs, 0)
    tr.check_device_status()
    prof = tr.profile_read()
    tr.profile_enable(0)
    assert prof["bwd_pipe_camera"][1] == 0 and prof["bwd_chain_camera"][1] == 1 and prof["wgrad_gemm"][1] == 1, prof   # no pipelined launch any more
    assert not torch.equal(tr.flat.detach(), p0)
    # a healthy field next to it still trains on the pipelined path
    l, g, _ = _grads(_field(True), 512, 0)
    assert torch.isfinite(g).all()


@pytest.mark.parametrize("epoch", [0, 3])
def test_status_check_between_forward_and_backward_keeps_the_backward_on_its_forwards_path(epoch):
    """eonerf_device_status switches a context off the pipelined path after a reported fault.  A backward whose forward ran
    BEFORE the switch (autograd paths keep a workspace across host code) must still run on the path its workspace was carved and its
    mask slots were written for; the context itself stays switched for the next forward."""
    import ctypes as C
    from eonerf_code_amd import _lib
    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.trainer import FusedTrainer
    R = 512
    f = _field(True)
    l_ref, g_ref, tr = _grads(f, R, epoch)
    rays, img, rgbs = (t.cuda() for t in synthetic_batch(R, N_IMG, seed=3))
    g = torch.Generator(device="cuda").manual_seed(3)
    noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
    flags = _lib.F_TRAIN | (_lib.F_SHADOWS if epoch >= 2 else _lib.F_RGB_LOSS)
    tr._render_forward(rays, img, R, flags, noise)
    tr.loss_grad(tr.out[:R], rgbs, epoch, tr.d_out)
    # a REMOTE rank's fault arrives (the reduced flag of the gradient message): k_adam skips the update and raises the sticky status word
    one = torch.ones(1, device="cuda")
    scratch = torch.zeros_like(tr.d_flat)
    _lib.check(tr.L.eonerf_adam_step(tr.ctx, C.c_void_p(tr.flat.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_void_p(tr.exp_avg.data_ptr()),
                                     C.c_void_p(tr.exp_avg_sq.data_ptr()), 1, 0.0, 0.9, 0.999, 1e-8, 1.0, C.c_void_p(one.data_ptr()), None))
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        tr.check_device_status()                     # reported: the context leaves the pipelined path ...
    tr._render_backward(rays, img, R, flags)         # ... but THIS backward belongs to a forward of the pipelined path
    tr.check_device_status()
    assert (tr.d_flat - g_ref).norm().item() <= 2e-6 * g_ref.norm().item()
    # the next step runs chain + GEMM end to end and agrees to summation order
    tr.profile_enable(2)
    tr.forward_backward(rays, img, rgbs, epoch, noise)
    tr.check_device_status()
    prof = tr.profile_read()
    tr.profile_enable(0)
    assert prof["bwd_pipe_camera"][1] == 0 and prof["wgrad_gemm"][1] == 1, prof
    assert (tr.d_flat - g_ref).norm().item() <= 1e-4 * g_ref.norm().item()


@pytest.mark.parametrize("epoch", [0, 3])
def test_pipelined_path_with_empty_rays_and_an_all_empty_batch(epoch):
    """What filter_pts_outside_cube (sat_rendering.py:18-22) can leave: a batch in which every second ray keeps no sample, and a batch
    with no sample at all (zero-step pipelines, zero-step GEMM items).  The launch drains, the status is clean, the gradients are
    finite -- all zero for the empty batch -- and the mixed batch agrees with the chain + GEMM path."""
    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.trainer import FusedTrainer
    R = 256
    rays, img, rgbs = (t.cuda() for t in synthetic_batch(R, N_IMG, seed=11))
    g = torch.Generator(device="cuda").manual_seed(11)
    noise = tuple(torch.rand(R, 128, device="cuda", generator=g) for _ in range(3))
    outside = rays.clone()
    outside[:, 0] = 5.0                              # origins far outside the cube: every sample is filtered
    mixed = rays.clone()
    mixed[::2] = outside[::2]
    grads = {}
    for pipe in (False, True):
        f = _field(pipe)
        tr = FusedTrainer(f, lr=0.0, max_rays=R)
        for tag, rr in (("mixed", mixed), ("all_empty", outside)):
            loss = float(tr.step(rr.contiguous(), img, rgbs, epoch, noise=noise))
            tr.check_device_status()
            assert loss == loss and torch.isfinite(tr.d_flat).all(), (pipe, tag)
            if tag == "all_empty":
                assert int(tr.n_samples.item()) == 0
                mlp = [v for (n, _), v in zip(f.named_parameters(), f.grad_views(tr.d_flat)) if "mlp" in n or "layer" in n]
                assert all(float(v.abs().max()) == 0.0 for v in mlp), "no sample: no gradient reaches the field"
            grads[(pipe, tag)] = tr.d_flat.clone()
    a, b = grads[(False, "mixed")], grads[(True, "mixed")]
    assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10


def test_four_times_the_bench_batch_and_the_size_guard():
    """16384 rays x 128 samples (2.1 M samples: the training slabs pass 12 GB, byte offsets inside them pass 2^32; every 256-row block is
    addressed through its own descriptor): pipelined path against chain + GEMM.  Beyond 66,050 rays a 256-row block would outgrow the
    32-bit offsets of its descriptor: the C ABI refuses (EONERF_E_UNSUPPORTED) instead of wrapping around."""
    import ctypes as C
    from eonerf_code_amd import _lib
    R = 16384
    f_old, f_new = _field(False), _field(True)
    l0, g0, _ = _grads(f_old, R, 3)
    torch.cuda.empty_cache()
    l1, g1, tr = _grads(f_new, R, 3)
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and torch.isfinite(g1).all()
    for (name, p), a, b in zip(f_old.named_parameters(), f_old.grad_views(g0), f_new.grad_views(g1)):
        assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10, (name, (a - b).norm().item(), a.norm().item())
    L = _lib.lib()
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    big = 70000
    dummy = torch.zeros(16, device="cuda")
    rc = L.eonerf_render_forward(tr.ctx, C.c_void_p(tr.flat.data_ptr()), C.c_void_p(dummy.data_ptr()), C.c_void_p(dummy.data_ptr()),
                                 C.c_void_p(tr.zsteps.data_ptr()), None, None, None, big, _lib.F_TRAIN | _lib.F_RGB_LOSS,
                                 C.c_void_p(dummy.data_ptr()), None, C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), None)
    assert rc == -4, rc                                   # EONERF_E_UNSUPPORTED, before anything is launched


@pytest.mark.parametrize("R,epoch", [(4096, 3), (4096, 0), (37, 3)])
def test_gemm_riders_match_the_jobs_they_replace(R, epoch, monkeypatch):
    """The sigma row and the transient head's embedding columns ride on the bottleneck-factor job of the weight-gradient GEMM
    (WgradAux, csrc/eonerf_wgrad.hip) instead of being jobs of their own (EONERF_WGRAD_RIDERS=0): the same products on the same
    operands, summed in another order -- every tensor to 1e-4, the two that the riders produce looked at by name."""
    monkeypatch.setenv("EONERF_WGRAD_RIDERS", "0")
    f_jobs = _field(True, seed=11)
    monkeypatch.delenv("EONERF_WGRAD_RIDERS")
    f_riders = _field(True, seed=11)
    l0, g0, _ = _grads(f_jobs, R, epoch)
    l1, g1, _ = _grads(f_riders, R, epoch)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    seen = set()
    for (name, _), a, b in zip(f_jobs.named_parameters(), f_jobs.grad_views(g0), f_riders.grad_views(g1)):
        assert (a - b).norm().item() <= 1e-4 * a.norm().item() + 1e-10, name
        if name in ("sigma_layer.output_layer.weight", "sigma_layer.output_layer.bias", "transient_mlp.hidden_layers.0.weight"):
            seen.add(name)
            if epoch >= 2 or "sigma" in name:
                assert a.norm().item() > 0, name
    assert len(seen) == 3


def test_deterministic_backward_repeats_bit_for_bit_many_times(monkeypatch):
    """Round 5: EONERF_DETERMINISTIC's weight-gradient reduce used to run one block per (job, row) with a plain "+=", although the camera
    pass' and the shadow pass' jobs of one layer write the same elements: about once in 1,500 backward passes (1024 rays) two blocks raced
    and a contribution to layer 0's gradient was lost (scripts/det_race_probe.py, profiles/r05_det_reduce_race.txt).  One forward, the
    backward 1,500 times into a zeroed buffer: every repetition reproduces the first one bit for bit."""
    from eonerf_code_amd import _lib
    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.trainer import FusedTrainer
    monkeypatch.setenv("EONERF_DETERMINISTIC", "1")
    R = 1024
    f = _field(True, seed=13)
    tr = FusedTrainer(f, lr=0.0, max_rays=R)
    rays, img, pix = (t.cuda() for t in synthetic_batch(R, N_IMG, seed=5))
    flags = _lib.F_TRAIN | _lib.F_SHADOWS
    tr._render_forward(rays, img, R, flags, (None, None, None))
    ref, bad = None, 0
    for _ in range(1500):
        tr._grad_clean = False
        tr._render_backward(rays, img, R, flags, pixels=pix, kind=1)
        g = tr.d_flat[:tr.n_params]
        if ref is None:
            ref = g.clone()
        elif not torch.equal(g, ref):
            bad += 1
    tr.check_device_status()
    assert ref.abs().max().item() > 0 and bad == 0, bad
