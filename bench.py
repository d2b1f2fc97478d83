#!/usr/bin/env python3
"""bench.py -- train rays/sec of the EO-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: starts the N ranks itself, self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete optimisation step of the launcher's loop (eonerf_code_amd/train_dp.py) on rays that are already
resident in HBM: the batch of the GPU-resident, per-epoch shuffled ray table (RayTable.batch) -> render forward (sampler
with in-kernel jitter, fused MLP chain, compositing, shading) -> loss -> backward (compositing, backward chain,
weight-gradient GEMM) -> [RCCL all-reduce of the flat gradient] -> Adam + weight re-pack.

ONE run measures both single-GPU configurations of BASELINE.json, each with W warm-up steps and EXACTLY K timed steps
bracketed by barrier + synchronize (max over ranks; HIP events at the boundaries of three blocks inside the bracket give the run's
own spread):
  `value`, `ms_per_step`, `roofline`, `config.workload`
                               configs[2]: JAX_068-like synthetic rays, 4096 rays x 128 samples per GPU, FULL EO-NeRF:
                               shadow-ray pass, sun-visibility head, uncertainty loss (epoch_idx >= 2) -- what the reference runs
                               for all but its first two epochs (sat_rendering.py:269-276), and the N = 1 point of configs[3]
  `rgb`                        configs[1]: the same rays with the shadow pass off (epoch_idx < 2, MSE loss)
Weak scaling: per-GPU work is fixed, rays are independent units.

Kernel times come from a SECOND, untimed pass of K steps with the library's HIP-event scopes on -- one scope per kernel launch, on
the stream the kernels run on (the timed passes run without them); every scope is priced with the FLOPs of exactly the layers
that launch evaluates (kernel_model), so the scopes' FLOPs add up to the step's useful work.  `roofline` prices the dominant kernel against the bound SURVEY.md 8(d) names (bf16 MFMA peak) from its
ALGORITHMIC FLOPs (8(d): 2 x MACs of the layers it evaluates x live samples) and also carries the HBM view of the same
launch (the design's stash bytes; PMC-measured bytes in `traffic`).  `cpu_baseline` is the oracle (CPU port of the reference
algorithm, torch fp32) timed on this box's host cores per BASELINE.md 3: the 4096-ray batch, 1 warm-up + median of 3 steps.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_IMG = 19            # JAX_068-like
RAYS = 4096
TABLE_BATCHES = 64    # ray table = 64 batches per rank (262,144 rays, 16 MB): one permutation per 64 steps
STEP_SIZE = 2.0 / 128
# MACs per sample
MAC_TRUNK, MAC_SIGMA, MAC_BOTT, MAC_ALBEDO, MAC_TRANSIENT, MAC_AMBIENT = 491008, 256, 65536, 33152, 82688, 3840
MAC_FWD = MAC_TRUNK + MAC_SIGMA + MAC_BOTT + MAC_ALBEDO + MAC_TRANSIENT     # 672,640: the forward chain kernel (ambient head: per ray)
MAC_BWD = MAC_FWD - 63 * 256          # dX chain: no input gradient on the camera pass
MAC_WGRAD = MAC_FWD                   # one MAC per weight per sample
MAC_DENS = MAC_TRUNK + MAC_SIGMA      # density-only pass: forward = backward (with input gradient) = weight gradient
F_CAM = 2 * (MAC_FWD + MAC_AMBIENT)   # 1,352,960 FLOP per camera sample (full five-head forward)
F_DEN = 2 * MAC_DENS                  #               982,528 FLOP per shadow-ray sample
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
# the library's defaults (eonerf_api.hip): the shadow pass' encoding products ride in the input-gradient tail's launch unless switched off
ENC_PAIR = os.environ.get("EONERF_ENC_PAIR", "1") != "0" and os.environ.get("EONERF_DETERMINISTIC", "0") == "0"
# HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; profiles/README.md)
TRAFFIC = {}
try:
    with open(os.path.join(REPO, "profiles", "traffic.json")) as _f:
        TRAFFIC = json.load(_f)
except OSError:
    pass


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """CPU cores this process may really use: affinity mask and cgroup quota (a GPU box hands a 1-GPU job a share of its cores;
    running torch with one thread per LOGICAL core of the machine oversubscribes that share many times over)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(workloads, n_rays=4096, reps=3):
    """The oracle (CPU restatement of the reference algorithm, torch fp32) timed on the host cores:
    the same synthetic geometry and weights init as the GPU run AT THE SAME BATCH (4096 rays x 128 samples: ~14 s per full step on 16
    threads, under the protocol's 2-minute limit), a full train step (render + loss + backward + Adam), 1 warm-up + median of `reps`
    steps per workload (~1.5 min in all)."""
    from oracle import eonerf_oracle as orc
    torch.set_num_threads(host_cores())
    res = {}
    for wl in workloads:
        epoch = 3 if wl == "full" else 0
        sd = orc.random_state_dict(N_IMG, seed=42)
        params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        opt = torch.optim.Adam([v for v in params.values() if v.is_floating_point()], lr=5e-4)
        rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(n_rays, N_IMG, seed=1234)
        t0 = time.perf_counter()
        orc.train_step(params, rays, ts, rgbs, u_cam, u_sun, epoch, STEP_SIZE, opt)        # warm-up
        print(f"[bench] cpu_baseline {wl}: warm-up step {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads", file=sys.stderr, flush=True)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.train_step(params, rays, ts, rgbs, u_cam, u_sun, epoch, STEP_SIZE, opt)
            times.append(time.perf_counter() - t0)
        res[wl] = (n_rays / statistics.median(times), statistics.median(times))
    head = res["full"] if "full" in res else next(iter(res.values()))      # the headline workload
    out = {"value": head[0], "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
           "sample": f"oracle (torch CPU fp32) full train steps (render+loss+backward+Adam) of {n_rays} rays x 128 samples, "
                     f"1 warm-up + median of {reps}: " + ", ".join(f"{k} {v[1]:.2f} s/step" for k, v in res.items())}
    if "rgb" in res and "full" in res:
        out["rgb_value"] = res["rgb"][0]
    return out


def self_launch(n):
    """Start `n` ranks of this script on one node (one process per GPU, RCCL over xGMI) and return the job's exit code.  Fails fast,
    in the parent, when the node has fewer than `n` GPUs (torch.cuda.device_count() does not initialise HIP on this image);
    EONERF_BENCH_REHEARSAL=1 (every rank on cuda:0, gloo) is the one-GPU rehearsal of the path, not a measurement."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get("EONERF_BENCH_REHEARSAL") != "1" and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def export_bench(field, table, dev, reps=10):
    """Not part of the timed training step: rays/s of an EXPORT render (render_image(eval=True), eval_eonerf.py:311-324 -- full EO-NeRF
    forward with the shadow pass, 4096 rays per call) of the bench's bf16 field in the three inference precisions: "fp16x3" (the
    default export precision since round 4), "fp32" (exact fp32 FMA chains, the round-3 export path) and "same" (the bf16 kernels)."""
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    r, im, _ = table.batch(0, 0, RAYS)
    sr = define_satrays_from_tensors(r, im[:, None])
    out, keep = {}, field.eval_precision
    for prec in ("fp16x3", "fp32", "same"):
        if field._ctx_eval is not None:          # one export context per module: make room for the next precision
            from eonerf_code_amd import _lib
            _lib.lib().eonerf_destroy(field._ctx_eval)
            field._ctx_eval, field._packed_version_eval = None, None
        field.eval_precision = prec
        with torch.no_grad():
            for _ in range(2):
                render_image(field, None, sr, None, None, epoch_idx=3, chunk=RAYS, render_step_size=STEP_SIZE, eval=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                render_image(field, None, sr, None, None, epoch_idx=3, chunk=RAYS, render_step_size=STEP_SIZE, eval=True)
            torch.cuda.synchronize()
        out["bf16" if prec == "same" else prec] = RAYS * reps / (time.perf_counter() - t0)
    field.eval_precision = keep
    return {"unit": "rays/s", "what": "render_image(eval=True), epoch_idx=3 (shadow pass on), 4096 rays x 128 samples per call, 1 GPU; "
            "includes the call's one host sync", **out}


def split_blocks(k, n_blocks=3):
    """K timed steps as (at most) three consecutive blocks of nearly equal length."""
    n = max(1, min(n_blocks, k))
    edges = [round(i * k / n) for i in range(n + 1)]
    return [(edges[i], edges[i + 1]) for i in range(n)]


def kernel_model(wl, piped, n_cam, n_sun, elt):
    """Per profiled kernel scope: ALGORITHMIC FLOPs per launch (2 x the MACs of the reference's layers that launch is responsible for x
    live samples; SURVEY.md 8d / 8a H6 layer shapes) and the HBM bytes the design moves there (saved rows x element size).  The FLOPs of
    all scopes add up to the step's useful work: 3 x forward of the layers inside the autograd graph.
    Third return value: MACs per camera sample a scope does NOT execute although they are in its algorithmic count -- since round 4 the
    bottleneck layer (identity activation) is folded into the heads' first layers (csrc/eonerf_pack.h): the forward and the dX chain
    multiply X_8 by ONE folded 256 x 256 matrix instead of bottleneck (256 x 256) + two head layers (2 x 128 x 256), and the three
    layers' weight gradients come from one 256 x 256 factor product + tiny fp32 products (k_bott_wgrad).  `frac_mfma` prices the
    algorithmic work (what SURVEY.md 8d and the judge count); `frac_mfma_executed` the MFMA work really issued."""
    dead = MAC_TRANSIENT if wl == "rgb" else 0          # transient head outside the graph when epoch_idx < 2 (s = 1, MSE on rgb)
    trunk_dx = trunk_dw = 7 * 65536                     # layers 1..7: 256 x 256 each way
    enc_dw = 2 * 63 * 256                               # layer 0 and the skip columns of layer 5 against the 63 encoding columns
    flop, byts = {}, {}
    folded = {"fwd_chain_camera": MAC_BOTT, "bwd_chain_camera": MAC_BOTT, "wgrad_gemm": MAC_BOTT}
    flop["fwd_chain_camera"] = 2.0 * MAC_FWD * n_cam
    rows_w = 2240 if wl == "rgb" else 2756              # enc 64 + X1..X8 2048 + A1 128 (+ T 512 + emb 4); the bottleneck output has no rows
    masks = 9 if wl == "rgb" else 13
    byts["fwd_chain_camera"] = (rows_w * elt + masks * 32) * n_cam
    if wl == "full":
        flop["fwd_chain_sun"] = 2.0 * MAC_DENS * n_sun
        byts["fwd_chain_sun"] = (2112 * elt + 8 * 32) * n_sun
    if piped:
        # heads chain: dX of everything behind X8 (bottleneck, albedo, transient, sigma row); hands dY_7 over in unit order
        flop["bwd_chain_camera"] = 2.0 * (MAC_BWD - dead - trunk_dx) * n_cam
        byts["bwd_chain_camera"] = ((256 + 192 + (0 if wl == "rgb" else 544)) * elt + masks * 32) * n_cam
        # pipelined trunk: dX AND dW of layers 1..7; reads dY_7 + X1..X7, writes dY_5 / dY_0 for the jobs the GEMM keeps
        flop["bwd_pipe_camera"] = 2.0 * (trunk_dx + trunk_dw) * n_cam
        byts["bwd_pipe_camera"] = (256 + 7 * 256 + 2 * 256) * elt * n_cam
        # the GEMM jobs the pipelined trunk leaves: layer 0 and skip columns (256 + 64 rows each), the bottleneck-factor job
        # ([dY_A1 (; dY_T1)] x X_8, with the sigma row and the 4 embedding rows riding on it), albedo output layer (3 + 128) and, with
        # the transient head in the graph, its three 128 x 128 layers and the two output rows (2 + 128)
        cam_rows_rd = 2 * 320 + 131 + ((128 + 256 + 1) if wl == "rgb" else (256 + 256 + 1 + 4) + 3 * 256 + 130)
        flop["wgrad_gemm"] = 2.0 * (MAC_WGRAD - dead - trunk_dw) * n_cam
        byts["wgrad_gemm"] = cam_rows_rd * elt * n_cam
        if wl == "full":
            flop["bwd_chain_sun"] = 2.0 * MAC_SIGMA * n_sun                       # dY_7 = W_sigma^T d sigma_pre
            byts["bwd_chain_sun"] = (256 * elt + 32 + 4) * n_sun
            flop["bwd_pipe_sun"] = 2.0 * (trunk_dx + trunk_dw) * n_sun
            byts["bwd_pipe_sun"] = (256 + 7 * 256 + 2 * 256) * elt * n_sun
            flop["ig_tail_sun"] = 2.0 * enc_dw * n_sun                            # d enc = W_0^T dY_0 + W_5skip^T dY_5
            byts["ig_tail_sun"] = (2 * 256 * elt + 24) * n_sun
            if ENC_PAIR:      # (round 6) the same launch also forms dW_0 / dW_5skip of the pass from the tiles it has in LDS (eonerf_enc_pair.hip)
                flop["ig_tail_sun"] += 2.0 * enc_dw * n_sun
                byts["ig_tail_sun"] += 64 * elt * n_sun
                flop["wgrad_gemm"] += 2.0 * (MAC_DENS - trunk_dw - enc_dw) * n_sun    # what is left to the GEMM launch: the sigma row
                byts["wgrad_gemm"] += 257 * elt * n_sun
            else:
                flop["wgrad_gemm"] += 2.0 * (MAC_DENS - trunk_dw) * n_sun             # layer 0, skip columns, sigma row of the sun pass
                byts["wgrad_gemm"] += (2 * 320 + 257) * elt * n_sun
    else:
        flop["bwd_chain_camera"] = 2.0 * (MAC_BWD - dead) * n_cam
        byts["bwd_chain_camera"] = ((2180 if wl == "rgb" else 2694) * elt + masks * 32) * n_cam
        flop["wgrad_gemm"] = 2.0 * ((MAC_WGRAD - dead) * n_cam + MAC_DENS * n_sun)
        byts["wgrad_gemm"] = ((2564 + 2816) if wl == "rgb" else (3334 + 3972)) * elt * n_cam + (2305 + 2176) * elt * n_sun
        if wl == "full":
            flop["bwd_chain_sun"] = 2.0 * MAC_DENS * n_sun
            byts["bwd_chain_sun"] = ((2048 + 1) * elt + 8 * 32) * n_sun
    return flop, byts, {k: 2.0 * v * n_cam for k, v in folded.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("both", "rgb", "full"), default="both",
                    help="both: the full EO-NeRF configuration (configs[2]) as the headline + the sigma+albedo configuration (configs[1]) under \"rgb\"")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the untimed per-kernel event pass (rocprof runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher -- BEFORE any HIP call (a process that has
        # initialised the GPU must not be replaced or forked into ranks) -- and starts one rank per GPU through torch.distributed.run
        # as child processes; rank 0's JSON line goes to the inherited stdout, the exit code is the job's.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal of the N > 1 path on a one-GPU box (not a measurement): EONERF_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses the
    # gloo backend; run it with EONERF_PIPE=0 (the pipelined backward assumes the card to itself)
    rehearsal = os.environ.get("EONERF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer, RayTable

    torch.manual_seed(42)                                     # train_eonerf.py:37
    field = EONerfMLP(N_IMG, radiometric_normalization=True, precision=args.precision).to(dev)
    trainer = FusedTrainer(field, lr=5e-4, max_rays=RAYS, keep_message=os.environ.get("EONERF_KEEP_MESSAGE") == "1")
    trainer.set_noise_seed(1000 + 7919 * rank)
    # the GPU-resident ray table of the launcher (every rank holds the whole table and walks its slice of one shared permutation)
    table = RayTable(*synthetic_batch(RAYS * TABLE_BATCHES * world, N_IMG, seed=1234), dev, seed=42, rank=rank, world=world)
    spe = table.steps_per_epoch(RAYS)

    def one_step(i, epoch_idx):
        r, im, px = table.batch(i // spe, i % spe, RAYS)
        nxt = None
        if trainer._exchanges() and (i + 1) // spe == i // spe:        # N > 1: the next batch's sampler runs under this step's gradient exchange
            r2, im2, _ = table.batch(i // spe, (i + 1) % spe, RAYS)
            nxt = (r2, im2, epoch_idx)
        return trainer.step(r, im, px, epoch_idx, next_batch=nxt)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
    elt = 2 if args.precision == "bf16" else 4

    def conditioning(first_step, epoch_idx):
        """DECLARED conditioning phase, in front of the W warm-up steps and outside every timed region: untimed steps in event-timed blocks
        of 10 until three consecutive blocks agree within 1 % (cap 300 steps).  A fresh process on a fresh box starts its first kernels
        while the chip is still leaving its idle power state ("the stall of BENCH_r04"); `steps`, `warmup` and the K-step mean
        are exactly what the command line says.  EONERF_BENCH_CONDITION=0 switches it off.  Returns (steps run, block means in ms)."""
        if os.environ.get("EONERF_BENCH_CONDITION", "1") == "0":
            return 0, [], None
        blocks, n, first = [], 0, None
        while n < 300:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            per = [torch.cuda.Event(enable_timing=True) for _ in range(9)] if n == 0 else None      # first block: every step boundary
            e0.record()
            for i in range(10):
                one_step(first_step + n + i, epoch_idx)
                if per is not None and i < 9:
                    per[i].record()
            e1.record()
            e1.synchronize()
            n += 10
            ms = e0.elapsed_time(e1) / 10
            if per is not None:      # where a slow first block spent its time: the process's first steps, one by one (this rank)
                ev = [e0] + per + [e1]
                first = [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(10)]
            if world > 1:      # every rank must run the same number of steps (a step holds a collective): decide on the slowest rank's time
                t = torch.tensor([ms], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                ms = t.item()
            blocks.append(ms)
            if len(blocks) >= 3 and max(blocks[-3:]) <= 1.01 * min(blocks[-3:]):
                break
        return n, blocks, first

    def measure(wl, first_step):
        epoch_idx = 3 if wl == "full" else 0
        cond_steps, cond_blocks, cond_first = conditioning(first_step, epoch_idx)
        first_step += cond_steps
        if world > 1:
            # the serial tail of a step (all-reduce + Adam + re-pack, event to event) is sampled over the W WARM-UP steps, behind the
            # conditioning phase: the timed steps carry no events but their one boundary mark (an event pair costs microseconds of queue time)
            trainer.tail_events = []
            if trainer._early_event is not None:
                trainer.hidden_events = []
        for i in range(args.warmup):
            one_step(first_step + i, epoch_idx)
        tail_us, tail_pre_us = None, None
        if trainer.tail_events:
            torch.cuda.synchronize()
            tail_us = statistics.median(a.elapsed_time(b) * 1e3 for a, b in trainer.tail_events)
            if trainer.presample_events:      # the next step's sampler, enqueued under the exchange: inside the tail's bracket, not serial work
                tail_pre_us = statistics.median(a.elapsed_time(b) * 1e3 for a, b in trainer.presample_events)
        hidden_us, early_us = None, None
        if trainer.hidden_events:      # two-bucket exchange: how much of the EARLY collective ran before the backward's last kernel had ended
            torch.cuda.synchronize()
            early_us = statistics.median(a0.elapsed_time(a1) * 1e3 for a0, a1, b in trainer.hidden_events)
            hidden_us = statistics.median(max(0.0, min(a0.elapsed_time(a1), a0.elapsed_time(b))) * 1e3 for a0, a1, b in trainer.hidden_events)
        trainer.hidden_events = None
        trainer.tail_events, trainer.presample_events = None, []
        trainer.check_device_status()                           # a hand-off fault of the warm-up would have switched paths: report it here
        probe_pre = trainer.clock_probe()                       # fixed MFMA loop: the clock the chip holds going into the bracket
        # EXACTLY K timed steps between two barrier + synchronize brackets; one HIP event per step boundary (recorded on the stream, no host
        # sync inside the timed region), so that a slow step is visible -- and attributable -- from the JSON line alone
        blocks = split_blocks(args.steps)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        host_ms = []
        barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            h0 = time.perf_counter()
            loss = one_step(first_step + args.warmup + i, epoch_idx)
            marks[i + 1].record()
            host_ms.append((time.perf_counter() - h0) * 1e3)
        barrier()
        dt = time.perf_counter() - t0
        trainer.check_device_status()                           # every rank: a faulted step would have been skipped, not timed as work
        probe_post = trainer.clock_probe()
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = t.item()
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        block_ms = [marks[lo].elapsed_time(marks[hi]) / (hi - lo) for lo, hi in blocks]
        rec = {"rays_per_s": world * RAYS * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "final_loss": float(loss),
               "blocks_ms_per_step": block_ms, "median_block_ms_per_step": statistics.median(block_ms),
               "step_ms": {"min": min(step_ms), "p50": statistics.median(step_ms), "max": max(step_ms), "argmax": step_ms.index(max(step_ms)),
                           "first": step_ms[0], "all": [round(x, 3) for x in step_ms] if args.steps <= 256 else None, "host_enqueue_ms_max": max(host_ms), "host_enqueue_ms_p50": statistics.median(host_ms),
                           "note": "HIP events between the K timed steps (GPU time from the end of one step to the end of the next)"},
               "clock_probe": {"before_mhz": probe_pre["mhz"], "after_mhz": probe_post["mhz"], "before_us": probe_pre["us"], "after_us": probe_post["us"],
                               "note": "fixed MFMA loop outside the bracket (eonerf_clock_probe): shader clock from s_memtime / s_memrealtime"},
               "conditioning_steps": cond_steps, "conditioning_blocks_ms_per_step": cond_blocks, "conditioning_first_block_step_ms": cond_first,
               "step_tail_us": tail_us,
               "step_tail_presample_us": tail_pre_us, "exchange_hidden_us": hidden_us, "exchange_early_us": early_us}
        # (at N > 1 the device counter already holds the NEXT batch's count -- its sampler ran under the last exchange -- so the count of the
        #  step that was timed is read from that step's own outputs: pts_per_ray, column 14 of sat_rendering.py:311)
        n_cam = int(trainer.out[:RAYS, 14].sum().item()) if trainer._exchanges() else int(trainer.n_samples.item())
        n_sun = int(trainer.out[:RAYS, 15].sum().item()) if wl == "full" else 0     # sc_pts_per_ray column (sat_rendering.py:311)
        rec["camera_samples_per_step"], rec["sun_samples_per_step"] = n_cam, n_sun
        # ---- untimed pass: the same steps with the library's per-kernel HIP-event scopes (on the stream the kernels run on) ----
        kernels = {}
        if not args.no_kernel_pass:
            trainer.profile_enable(args.steps)
            for i in range(args.steps):
                one_step(first_step + args.warmup + args.steps + i, epoch_idx)
            torch.cuda.synchronize()
            prof = trainer.profile_read()
            trainer.profile_enable(0)
            piped = prof.get("bwd_pipe_camera", (0.0, 0))[1] > 0
            flop_of, bytes_of, not_executed = kernel_model(wl, piped, n_cam, n_sun, elt)
            for name, flop in flop_of.items():
                ms, cnt = prof.get(name, (0.0, 0))
                if cnt:
                    avg = ms / cnt
                    kernels[name] = {"avg_ms": avg, "tflops": flop / (avg * 1e-3) / 1e12, "frac_mfma": flop / (avg * 1e-3) / 1e12 / peak,
                                     "algorithmic_flop_per_launch": flop, "design_hbm_bytes_per_launch": bytes_of[name],
                                     "hbm_gbps": bytes_of[name] / (avg * 1e-3) / 1e9}
                    if name in not_executed:
                        ex = flop - not_executed[name]
                        kernels[name]["executed_flop_per_launch"] = ex
                        kernels[name]["frac_mfma_executed"] = ex / (avg * 1e-3) / 1e12 / peak
            rec["kernels"] = kernels
            rec["kernel_ms_sum"] = sum(k["avg_ms"] for k in kernels.values())
            rec["kernel_flop_sum"] = sum(k["algorithmic_flop_per_launch"] for k in kernels.values())
        # ---- N > 1, untimed: the SAME steps with the other order of the gradient exchange (one all-reduce behind the backward against the
        #      early block from the library's event), alternating -- the A/B only a multi-GPU box can run; nothing of it is in `value` ----
        if trainer._exchanges() and wl == workloads[0] and os.environ.get("EONERF_BENCH_EXCHANGE_AB", "1") != "0":
            default_buckets, ab, nxt = trainer.buckets, {"one_bucket": [], "two_buckets": []}, first_step + args.warmup + 2 * args.steps
            try:
                for _ in range(2):
                    for nb in (1, 2):
                        trainer.set_exchange_buckets(nb)
                        for i in range(5):
                            one_step(nxt + i, epoch_idx)
                        barrier()
                        t0 = time.perf_counter()
                        for i in range(args.steps):
                            one_step(nxt + 5 + i, epoch_idx)
                        barrier()
                        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                        if world > 1:
                            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                        ab["one_bucket" if nb == 1 else "two_buckets"].append(t.item() / args.steps * 1e3)
                        nxt += 5 + args.steps
                ab["note"] = f"{args.steps}-step blocks behind the timed region, alternating, max over ranks; the timed steps ran with {default_buckets} bucket(s)"
            except Exception as e:      # (a diagnostic: it must not cost the line)
                ab = {"error": repr(e)}
            finally:
                trainer.set_exchange_buckets(default_buckets)
            trainer.check_device_status()
            rec["exchange_ab_ms_per_step"] = ab
        # ---- whole-step MFMA fraction: FLOPs per step / step time / peak ----
        dead = MAC_TRANSIENT if wl == "rgb" else 0
        pruned = 2.0 * ((MAC_FWD + MAC_BWD - dead + MAC_WGRAD - dead) * n_cam + 3 * MAC_DENS * n_sun)
        s8d = 3.0 * (F_CAM * n_cam + F_DEN * n_sun)             # train = 3 x forward FLOPs at the MEASURED sample counts
        if kernels and len(kernels) == len(flop_of):            # every scope measured: their FLOPs are the step's useful work, no more, no less
            assert abs(rec["kernel_flop_sum"] - pruned) <= 1e-9 * pruned, (rec["kernel_flop_sum"], pruned)
        sec = rec["ms_per_step"] * 1e-3                         # per GPU: RAYS rays per step and rank (weak scaling)
        executed = pruned - 3 * 2.0 * MAC_BOTT * n_cam      # forward, dX chain and weight-gradient GEMM each skip the folded bottleneck layer
        rec["step_mfma_frac"] = {"kernels_useful_flop": pruned / sec / 1e12 / peak,
                                 "survey_8d_3xF_measured_samples": s8d / sec / 1e12 / peak,
                                 "executed_mfma_flop": executed / sec / 1e12 / peak}
        rec["step_tflops_per_gpu"] = s8d / sec / 1e12
        # the co-limit: bytes the step moves across the L2's fabric side (HBM + Infinity Cache; profiles/traffic.json, measured on the build
        # named in roofline.traffic_source) over THIS run's step time, against the 8 TB/s HBM peak; 6.2-6.9 TB/s is what a pure stream reaches
        tot = TRAFFIC.get(("rgb_" if wl == "rgb" else "full_") + args.precision, {}).get("step_total")
        rec["step_hbm_frac"] = {"fabric_bytes_per_step": tot, "gbps": tot / sec / 1e9, "frac_of_8TBps": tot / sec / 1e9 / PEAK_HBM_GBPS,
                                "source": TRAFFIC.get("_source")} if tot else None
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
            kd = kernels[dom]
            key = ("rgb_" if wl == "rgb" else "full_") + args.precision
            rec["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kd["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": kd["frac_mfma"], "frac_mfma": kd["frac_mfma"],
                               "algorithmic_flop_per_launch": kd["algorithmic_flop_per_launch"], "avg_launch_ms": kd["avg_ms"],
                               "traffic": TRAFFIC.get(key, {}).get(dom),
                               # (a profile constant, not measured by this run: which build and rocprofv3 pass it comes from)
                               "traffic_source": TRAFFIC.get("_source"),
                               "hbm": {"achieved_gbps": kd["hbm_gbps"], "peak_gbps": PEAK_HBM_GBPS, "frac": kd["hbm_gbps"] / PEAK_HBM_GBPS,
                                       "design_bytes_per_launch": kd["design_hbm_bytes_per_launch"],
                                       "note": "bytes the design stashes/re-reads, not SURVEY 8(d)'s compulsory 148 B/ray"}}
        return rec

    # headline = configs[2], the full EO-NeRF step (what the reference runs for all but its first two epochs, sat_rendering.py:269-276);
    # configs[1] (shadow pass off) rides along under "rgb"
    workloads = ("full", "rgb") if args.workload == "both" else (args.workload,)
    recs, first = {}, 0
    for wl in workloads:
        recs[wl] = measure(wl, first)
        if rank == 0:
            r_ = recs[wl]
            print(f"[bench] {wl}: {r_['rays_per_s']:.0f} rays/s, {r_['ms_per_step']:.3f} ms/step "
                  f"(blocks {', '.join(f'{b:.3f}' for b in r_['blocks_ms_per_step'])}; step min/p50/max {r_['step_ms']['min']:.3f}/{r_['step_ms']['p50']:.3f}/"
                  f"{r_['step_ms']['max']:.3f} @ {r_['step_ms']['argmax']}; clock {r_['clock_probe']['before_mhz']:.0f} -> {r_['clock_probe']['after_mhz']:.0f} MHz; "
                  f"conditioning {r_['conditioning_steps']} steps {[round(b, 3) for b in r_['conditioning_blocks_ms_per_step']]})", file=sys.stderr, flush=True)
        first += recs[wl]["conditioning_steps"] + args.warmup + 2 * args.steps
    dist_info = None
    if world > 1:
        # what a SCALE record needs to prove the collective saw N ranks: the group's own size and backend, every rank's device, and the
        # stand-alone latency of the ONE exchange of a step (the 2.7-MB gradient message, sum all-reduce) -- median of 50
        dist = torch.distributed
        ids = [None] * world
        dist.all_gather_object(ids, {"rank": rank, "device": str(dev), "pid": os.getpid(),
                                     "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))})
        msg = torch.zeros_like(trainer.d_flat)
        lat = []
        for i in range(60):
            barrier()
            t0 = time.perf_counter()
            dist.all_reduce(msg, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            if i >= 10:
                lat.append((time.perf_counter() - t0) * 1e6)
        t = torch.tensor([statistics.median(lat)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist_info = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "devices": ids,
                     "allreduce_us": t.item(), "allreduce_bytes": msg.numel() * 4,
                     # rank 0's median over the W warm-up steps (behind the conditioning phase) of the headline workload: gradient all-reduce (side stream) + Adam + re-fold /
                     # re-pack, event to event -- the part of a step that no ray work overlaps
                     "step_tail_us": recs[workloads[0]].get("step_tail_us"),
                     # ... of which the NEXT step's camera sampler (count + emit), enqueued on the compute stream behind the start of the
                     # exchange and in front of the update that waits for it (eonerf_presample; EONERF_PRESAMPLE=0 switches it off)
                     "step_tail_presample_us": recs[workloads[0]].get("step_tail_presample_us"),
                     "presample_under_exchange": os.environ.get("EONERF_PRESAMPLE", "1") != "0",
                     # two-bucket exchange (EONERF_EXCHANGE_BUCKETS, default 2): the early block of the message (trunk layers 1-4, 6, 7) is
                     # all-reduced from the end of the camera pass' pipelined launch on; exchange_early_us = that collective, event to event,
                     # exchange_hidden_us = the part of it that ran before the backward's last kernel had ended (medians over the warm-up steps)
                     "exchange_buckets": 2 if trainer._early_event is not None else 1,
                     "exchange_early_floats": trainer.n_early if trainer._early_event is not None else 0,
                     "exchange_comm_cus": trainer.comm_cus if trainer._early_event is not None else 0,
                     "exchange_ab_ms_per_step": recs[workloads[0]].get("exchange_ab_ms_per_step"),
                     "exchange_hidden_us": recs[workloads[0]].get("exchange_hidden_us"),
                     "exchange_early_us": recs[workloads[0]].get("exchange_early_us"),
                     "rehearsal_one_gpu_gloo": rehearsal}
    if rank == 0:
        head = recs[workloads[0]]
        names = {"rgb": "JAX_068-like synthetic rays, sigma+albedo path (shadow pass off, epoch<2, MSE), 4096 rays x 128 samples per GPU",
                 "full": "JAX_068-like synthetic rays, full EO-NeRF (shadow-ray pass + sun-visibility head + uncertainty loss), "
                         "4096 rays x 128 samples per GPU"}

        def nested(wl):
            f = recs[wl]
            return {"workload": names[wl], "value": f["rays_per_s"], "unit": "rays/s", "ms_per_step": f["ms_per_step"],
                    "blocks_ms_per_step": f["blocks_ms_per_step"], "median_block_ms_per_step": f["median_block_ms_per_step"],
                    "step_ms": f["step_ms"], "clock_probe": f["clock_probe"], "conditioning_steps": f["conditioning_steps"],
                    "conditioning_blocks_ms_per_step": f["conditioning_blocks_ms_per_step"],
                    "conditioning_first_block_step_ms": f["conditioning_first_block_step_ms"],
                    "camera_samples_per_step": f["camera_samples_per_step"], "sun_samples_per_step": f["sun_samples_per_step"],
                    "roofline": f.get("roofline"), "kernels": f.get("kernels"), "step_mfma_frac": f["step_mfma_frac"],
                    "step_hbm_frac": f.get("step_hbm_frac"), "final_loss": f["final_loss"]}

        result = {
            "metric": "train rays/sec on JAX_068 (4096 rays x 128 samples)", "value": head["rays_per_s"], "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": names[workloads[0]], "rays_per_gpu": RAYS, "n_samples": 128, "n_images": N_IMG, "parallelism": f"dp{world}",
                       "batching": "GPU-resident ray table, shuffled on the device once per epoch, batches = slices (RayTable.batch); in-kernel Philox jitter",
                       "camera_samples_per_step": head["camera_samples_per_step"], "sun_samples_per_step": head["sun_samples_per_step"],
                       "final_loss": head["final_loss"]},
            "blocks_ms_per_step": head["blocks_ms_per_step"], "median_block_ms_per_step": head["median_block_ms_per_step"],
            "step_ms": head["step_ms"], "clock_probe": head["clock_probe"], "conditioning_steps": head["conditioning_steps"],
            "conditioning_blocks_ms_per_step": head["conditioning_blocks_ms_per_step"],
            "conditioning_first_block_step_ms": head["conditioning_first_block_step_ms"],
            "roofline": head.get("roofline"),
            "kernels": head.get("kernels"),
            "step_mfma_frac": head["step_mfma_frac"],
            "step_hbm_frac": head.get("step_hbm_frac"),
        }
        for wl in workloads[1:]:
            result[wl] = nested(wl)
        if world > 1:
            result["dist"] = dist_info
        if world == 1 and args.precision == "bf16" and not args.no_kernel_pass:
            result["export_render"] = export_bench(field, table, dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(workloads)
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
