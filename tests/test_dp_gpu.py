"""GPU, two ranks on ONE card (H11): FusedTrainer.step under torch.distributed -- the HIP backward's flat
gradient goes through trainer.reduce_gradients (side stream, Adam waits for it), here over gloo because both ranks share
cuda:0 (RCCL needs one device per rank; the driver's multi-GPU bench runs the same code path over RCCL).

  * both ranks end the step with bit-identical parameters (replicas stay identical without any parameter exchange);
  * the reduced gradient x 1/world equals the gradient of ONE process on the whole 2R batch (mean loss over 2R rays) to fp32
    reduction tolerance: 2e-4 relative L2 per tensor (split-K sums and atomics in a different order);
  * a non-zero C-ABI return code on one rank ends the job with a failure instead of hanging the peers.
The workers are fresh processes (started before they touch the GPU) and exit non-zero on any error.
"""
import os
import socket
import subprocess
import sys
import time

import pytest
import torch

from oracle import eonerf_oracle as orc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_IMG, R, WORLD = 4, 256, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_job(out_dir, mode, epoch, timeout=300, world=WORLD, kill_on_failure=True):
    """A miniature of what torchrun does: start one process per rank, fail the job when any rank fails."""
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), str(r), str(world), port, str(out_dir), mode,
                               str(epoch)], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    t0, codes = time.time(), [None] * world
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if (kill_on_failure and any(c not in (None, 0) for c in codes)) or time.time() - t0 > timeout:
            for p in procs:                      # exactly the PIDs started above
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    outs = [p.communicate()[0] for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("epoch", [0, 3])
def test_two_ranks_on_one_gpu_identical_replicas_and_global_batch_gradient(tmp_path, epoch):
    codes, outs = _run_job(tmp_path, "ok", epoch)
    assert codes == [0, 0], outs
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(WORLD))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    assert torch.equal(r0["d_flat"], r1["d_flat"]), "every rank must hold the same reduced gradient"
    # single process, whole 2R batch, rank 0's initial weights (the broadcast source)
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer
    sd = orc.random_state_dict(N_IMG, seed=91, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision="fp32")
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    tr = FusedTrainer(f, lr=5e-4, max_rays=R * WORLD)
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R * WORLD, N_IMG, seed=92)
    loss = tr.step(rays.cuda(), ts.reshape(-1).cuda(), rgbs.cuda(), epoch, noise=(u_cam.cuda(), None, u_sun.cuda()))
    ref = tr.d_flat.cpu()
    got = r0["d_flat"] / WORLD
    assert abs(float(loss) - 0.5 * (r0["loss"] + r1["loss"])) < 1e-5
    for (name, p), g_ref, g_got in zip(f.named_parameters(), f.grad_views(ref), f.grad_views(got)):
        assert (g_ref - g_got).norm().item() <= 2e-4 * g_ref.norm().item() + 1e-9, (epoch, name)
    # and the parameter update agrees (Adam's first step is lr * sign-like: compare where the gradient is not tiny)
    ref = ref[:tr.n_params]                                            # (the message's 4 control floats are not parameters)
    big = ref.abs() > 1e-4 * ref.abs().max()
    assert (tr.flat.detach().cpu() - r0["flat"])[big].abs().max().item() <= 2e-5


def test_nonzero_c_abi_return_code_on_one_rank_fails_the_job(tmp_path):
    codes, outs = _run_job(tmp_path, "fail", 0, timeout=120)
    assert codes[1] not in (0, None) and "libeonerf_hip" in outs[1], outs[1]
    assert not all(c == 0 for c in codes)
    assert not (tmp_path / "rank1.pt").exists()


def _single_process_reference(precision, epoch):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer
    sd = orc.random_state_dict(N_IMG, seed=91, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision=precision)
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    tr = FusedTrainer(f, lr=5e-4, max_rays=R * WORLD)
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R * WORLD, N_IMG, seed=92)
    loss = tr.step(rays.cuda(), ts.reshape(-1).cuda(), rgbs.cuda(), epoch, noise=(u_cam.cuda(), None, u_sun.cuda()))
    tr.check_device_status()
    return f, tr, float(loss)


@pytest.mark.parametrize("epoch", [0, 3])
def test_two_ranks_bf16_pipelined_backward_under_torch_distributed(tmp_path, epoch):
    """The benchmarked path (bf16, layer-pipelined backward) with torch.distributed initialised and a real gradient exchange: the
    ranks take turns on the card for the render/backward half, then all-reduce and update together."""
    codes, outs = _run_job(tmp_path, "pipe", epoch)
    assert codes == [0, 0], outs
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(WORLD))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    assert torch.equal(r0["d_flat"], r1["d_flat"])
    assert r0["d_flat"][-4].item() == 0.0                             # the fault flag of the reduced message
    f, tr, loss = _single_process_reference("bf16", epoch)
    ref, got = tr.d_flat.cpu(), r0["d_flat"] / WORLD
    assert abs(loss - 0.5 * (r0["loss"] + r1["loss"])) < 1e-5
    for (name, p), g_ref, g_got in zip(f.named_parameters(), f.grad_views(ref), f.grad_views(got)):
        assert (g_ref - g_got).norm().item() <= 2e-4 * g_ref.norm().item() + 1e-9, (epoch, name)


def test_device_side_fault_on_one_rank_stops_every_rank_before_the_update(tmp_path):
    """Rank 1's pipelined backward hits its watchdog.  The fault flag rides on the gradient all-reduce: neither rank's Adam kernel
    applies the (poisoned) update, the replicas stay identical, and BOTH ranks raise at their next status check."""
    codes, outs = _run_job(tmp_path, "fault", 0, timeout=180, kill_on_failure=False)
    assert all(c not in (0, None) for c in codes), (codes, outs)
    assert all("hand-off timed out" in o for o in outs), outs
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(WORLD))
    assert r0["d_flat"][-4].item() >= 1.0 and torch.equal(r0["d_flat"][-4:], r1["d_flat"][-4:])
    assert torch.equal(r0["flat"], r0["flat_before"]) and torch.equal(r1["flat"], r1["flat_before"])
    assert torch.equal(r0["flat"], r1["flat"])


def test_rccl_process_group_world_one_pipelined_step(tmp_path):
    """backend "nccl" (= RCCL): communicator creation and the gradient all-reduce on the side stream really execute (world size 1 is
    all a one-GPU box allows), next to the pipelined backward, in one process."""
    codes, outs = _run_job(tmp_path, "nccl1", 3, world=1)
    assert codes == [0], outs
    r0 = torch.load(tmp_path / "rank0.pt")
    f, tr, loss = _single_process_reference_world1(3)
    assert abs(loss - r0["loss"]) < 1e-5
    assert (tr.d_flat.cpu() - r0["d_flat"]).norm().item() <= 2e-4 * tr.d_flat.norm().item()


@pytest.mark.parametrize("epoch", [0, 3])
def test_next_steps_sampler_under_the_exchange_changes_nothing(tmp_path, epoch, monkeypatch):
    """FusedTrainer.step(next_batch=...) enqueues the next step's camera sampler (eonerf_presample) behind the start of the
    gradient all-reduce and in front of the Adam kernel that waits for it.  Five steps over RCCL (world 1, forced all-reduce, deterministic
    backward, in-kernel noise) with and without it: the same losses and the same parameters, bit for bit."""
    for mode in ("pre0", "pre1"):
        codes, outs = _run_job(tmp_path, mode, epoch, world=1)
        assert codes == [0], outs
    a, b = torch.load(tmp_path / "pre0.pt"), torch.load(tmp_path / "pre1.pt")
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    assert a["n_samples"] == b["n_samples"]
    assert torch.equal(a["flat"], b["flat"])
    if epoch == 3:      # the exchange as all_reduce(async_op=True) + wait() on the process group's stream (EONERF_EXCHANGE=async): the same again
        monkeypatch.setenv("EONERF_EXCHANGE", "async")
        codes, outs = _run_job(tmp_path, "pre1", epoch, world=1)
        assert codes == [0], outs
        c = torch.load(tmp_path / "pre1.pt")
        assert c["loss"] == a["loss"] and torch.equal(c["flat"], a["flat"])


def _single_process_reference_world1(epoch):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer
    sd = orc.random_state_dict(N_IMG, seed=91, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16")
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    tr = FusedTrainer(f, lr=5e-4, max_rays=R)
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R, N_IMG, seed=92)
    loss = tr.step(rays.cuda(), ts.reshape(-1).cuda(), rgbs.cuda(), epoch, noise=(u_cam.cuda(), None, u_sun.cuda()))
    return f, tr, float(loss)


def _torchrun(args, env_extra, timeout=600):
    port = str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port] + args
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_n_gt_1_path_rehearsal_two_ranks_on_one_card():
    """bench.py's N > 1 logic exactly as the driver launches it (torch.distributed.run, rendezvous from the environment, barrier-bracketed
    timing, MAX over ranks, rank 0's JSON line) -- rehearsed with two ranks on one card over gloo (EONERF_BENCH_REHEARSAL=1, chain + GEMM
    backward: two pipelined launches must not share a card).  Not a measurement: the checks are the line's shape and bookkeeping."""
    import json
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-pass"],
                  {"EONERF_BENCH_REHEARSAL": "1", "EONERF_PIPE": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak" and line["steps"] == 3
    assert "full EO-NeRF" in line["config"]["workload"]
    # whole-job rate = both ranks' rays over the slowest rank's time
    assert abs(line["value"] - 2 * 4096 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert "cpu_baseline" not in line                                  # rank 0 at N = 1 only


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher in the command (the form the driver's N = 1 record shows): the parent starts the two
    ranks itself (bench.self_launch) and relays rank 0's line, which carries the proof that the collective saw two ranks."""
    import json
    env = dict(os.environ, EONERF_BENCH_REHEARSAL="1", EONERF_PIPE="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-pass",
                        "--workload", "rgb"], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    d = line["dist"]
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["rehearsal_one_gpu_gloo"] is True
    assert sorted(x["rank"] for x in d["devices"]) == [0, 1] and len({x["pid"] for x in d["devices"]}) == 2
    assert d["allreduce_us"] > 0 and d["allreduce_bytes"] > 2_700_000
    assert d["step_tail_us"] > 0                                       # exchange + update + re-pack, event to event, median of the warm-up steps
    assert d["presample_under_exchange"] is True and 0 < d["step_tail_presample_us"] < d["step_tail_us"]      # the next step's sampler inside it
    assert line["conditioning_steps"] >= 20 and len(line["step_ms"]["all"]) == 2
    assert len(line["conditioning_first_block_step_ms"]) == 10 and min(line["conditioning_first_block_step_ms"]) > 0
    # the untimed A/B of the exchange's two orders behind the timed region: two blocks each, the default order restored afterwards
    ab = d["exchange_ab_ms_per_step"]
    assert "error" not in ab and len(ab["one_bucket"]) == 2 and len(ab["two_buckets"]) == 2 and min(ab["one_bucket"] + ab["two_buckets"]) > 0
    assert d["exchange_buckets"] == 2


def test_launcher_two_ranks_keeps_replicas_identical(tmp_path):
    """train_dp.py under torch.distributed.run with two ranks (one card, gloo rehearsal): broadcast at start, per-rank slices of one
    shared permutation, per-rank jitter seeds, one all-reduce per step -- the replicas end bit-identical."""
    r = _torchrun(["-m", "eonerf_code_amd.train_dp", "--synthetic_rays", "8192", "--batch_size", "512", "--n_images", "5",
                   "--max_train_steps", "20", "--check_every", "10", "--logs_dir", str(tmp_path), "--exp_name", "t",
                   "--dump_params", str(tmp_path / "params")],
                  {"EONERF_DP_REHEARSAL": "1", "EONERF_PIPE": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    p0, p1 = torch.load(str(tmp_path / "params") + ".rank0"), torch.load(str(tmp_path / "params") + ".rank1")
    assert torch.equal(p0, p1) and torch.isfinite(p0).all()
    assert "rays/s=" in r.stdout


def test_two_bucket_exchange_is_bit_identical_to_the_single_all_reduce(tmp_path, monkeypatch):
    """the gradient message goes out in two collectives by default -- [0, eonerf_grad_early_floats) (the trunk layers the camera
    pass' pipelined launch completes) from the library's exchange event on, the rest behind the last gradient kernel -- and
    EONERF_EXCHANGE_BUCKETS=1 gives the single all-reduce back.  With fixed-order gradient sums both give the same bits: two ranks over gloo
    on one card (pipelined bf16 step, shadow pass on), and five steps over RCCL at world size 1 with the forced collective."""
    res = {}
    for buckets in ("1", "2"):
        monkeypatch.setenv("EONERF_EXCHANGE_BUCKETS", buckets)
        d = tmp_path / f"gloo{buckets}"
        d.mkdir()
        codes, outs = _run_job(d, "pipedet", 3)
        assert codes == [0, 0], outs
        r0, r1 = (torch.load(d / f"rank{r}.pt") for r in range(WORLD))
        assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["d_flat"], r1["d_flat"])
        d1 = tmp_path / f"rccl{buckets}"
        d1.mkdir()
        codes, outs = _run_job(d1, "pre0", 3, world=1)
        assert codes == [0], outs
        res[buckets] = (r0, torch.load(d1 / "pre0.pt"))
    # ... and the order may change from step to step (FusedTrainer.set_exchange_buckets: what bench.py's untimed A/B at N > 1 does)
    dt = tmp_path / "rccl_toggle"
    dt.mkdir()
    codes, outs = _run_job(dt, "pretog", 3, world=1)
    assert codes == [0], outs
    tog = torch.load(dt / "pretog.pt")
    assert torch.equal(tog["flat"], res["2"][1]["flat"]) and tog["loss"] == res["2"][1]["loss"]
    (g1, n1), (g2, n2) = res["1"], res["2"]
    assert torch.equal(g1["d_flat"], g2["d_flat"]) and torch.equal(g1["flat"], g2["flat"]) and g1["loss"] == g2["loss"]
    assert torch.equal(n1["flat"], n2["flat"]) and n1["loss"] == n2["loss"]
    assert not torch.equal(g1["flat"], g1["flat_before"])                     # (a step was taken)


def test_exchange_event_fires_where_the_early_block_is_final():
    """The library records the exchange event behind the camera pass' pipelined launch: everything the later kernels of the backward (GEMM
    launch, tail) add to the gradient message lies at or beyond eonerf_grad_early_floats().  One backward with an event armed; a side stream
    that waits for the event copies the early block; the copy equals the block after the whole backward bit for bit, and the early block
    is exactly the trunk layers 1-4, 6, 7."""
    from eonerf_code_amd import _lib
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP, _ptr
    from eonerf_code_amd.trainer import FusedTrainer
    import ctypes as C
    sd = orc.random_state_dict(N_IMG, seed=91, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16")
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    tr = FusedTrainer(f, lr=5e-4, max_rays=4096)
    n_early = int(tr.L.eonerf_grad_early_floats(tr.ctx))
    names = sorted(n for n, off, r, c in f._layout if off < n_early)
    assert names == sorted(f"base_mlp.hidden_layers.{l}.{w}" for l in (1, 2, 3, 4, 6, 7) for w in ("weight", "bias"))
    assert n_early == 6 * (256 * 256 + 256)
    ev = torch.cuda.Event()
    ev.record()
    _lib.check(tr.L.eonerf_set_exchange_event(tr.ctx, C.c_void_p(ev.cuda_event), 8))
    rays, ts, rgbs, _, _ = orc.synthetic_batch(4096, N_IMG, seed=92)
    side = torch.cuda.Stream()
    tr.forward_backward(rays.cuda(), ts.reshape(-1).cuda(), rgbs.cuda(), 3)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        snap = tr.d_flat[:n_early].clone()
    torch.cuda.synchronize()
    tr.check_device_status()
    assert snap.abs().sum().item() > 0 and torch.equal(snap, tr.d_flat[:n_early])
    _lib.check(tr.L.eonerf_set_exchange_event(tr.ctx, None, 0))
