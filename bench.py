#!/usr/bin/env python3
"""bench.py -- train rays/sec of the EO-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete optimisation step of the launcher's loop (eonerf_code_amd/train_dp.py) on rays that are already
resident in HBM: the batch of the GPU-resident, per-epoch shuffled ray table (RayTable.batch) -> render forward (sampler
with in-kernel jitter, fused MLP chain, compositing, shading) -> loss -> backward (compositing, backward chain,
weight-gradient GEMM) -> [RCCL all-reduce of the flat gradient] -> Adam + weight re-pack.

ONE run measures both single-GPU configurations of BASELINE.json, each with W warm-up steps and EXACTLY K timed steps
bracketed by barrier + synchronize (max over ranks):
  `value`, `ms_per_step`, ...  configs[1]: JAX_068-like synthetic rays, 4096 rays x 128 samples per GPU,
                               shadow pass off (epoch_idx < 2, MSE loss), bf16 MFMA   <- the metric's configuration
  `full`                       configs[2]: the same rays with the shadow-ray pass, sun-visibility head and uncertainty loss
                               (epoch_idx >= 2) -- what the reference runs for all but its first two epochs
Weak scaling: per-GPU work is fixed, rays are independent units.

Kernel times come from a SECOND, untimed pass of K steps with the library's HIP-event brackets on (the timed passes run
without them).  `roofline` prices the dominant kernel against the bound SURVEY.md 8(d) names (bf16 MFMA peak) from its
ALGORITHMIC FLOPs (8(d): 2 x MACs of the layers it evaluates x live samples) and also carries the HBM view of the same
launch (the design's stash bytes; PMC-measured bytes in `traffic`).  `cpu_baseline` is the oracle (CPU port of the reference
algorithm, torch fp32) timed on this box's host cores per BASELINE.md 3: >= 1024 rays, 1 warm-up + median of 3 steps.
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


def cpu_baseline(workloads, n_rays=1024, reps=3):
    """The oracle (CPU restatement of the reference algorithm, torch fp32) timed on the host cores:
    the same synthetic geometry and weights init as the GPU run, a full train step (render + loss + backward + Adam),
    1 warm-up + median of `reps` steps per workload."""
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
    head = res["rgb"] if "rgb" in res else next(iter(res.values()))
    out = {"value": head[0], "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
           "sample": f"oracle (torch CPU fp32) full train steps (render+loss+backward+Adam) of {n_rays} rays x 128 samples, "
                     f"1 warm-up + median of {reps}: " + ", ".join(f"{k} {v[1]:.2f} s/step" for k, v in res.items())}
    if "full" in res:
        out["full_value"] = res["full"][0]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("both", "rgb", "full"), default="both",
                    help="both: the metric's configuration (rgb) as the headline + the full EO-NeRF configuration under \"full\"")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the untimed per-kernel event pass (rocprof runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    trainer = FusedTrainer(field, lr=5e-4, max_rays=RAYS)
    trainer.set_noise_seed(1000 + 7919 * rank)
    # the GPU-resident ray table of the launcher (every rank holds the whole table and walks its slice of one shared permutation)
    table = RayTable(*synthetic_batch(RAYS * TABLE_BATCHES * world, N_IMG, seed=1234), dev, seed=42, rank=rank, world=world)
    spe = table.steps_per_epoch(RAYS)

    def one_step(i, epoch_idx):
        r, im, px = table.batch(i // spe, i % spe, RAYS)
        return trainer.step(r, im, px, epoch_idx)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
    elt = 2 if args.precision == "bf16" else 4

    def measure(wl, first_step):
        epoch_idx = 3 if wl == "full" else 0
        for i in range(args.warmup):
            one_step(first_step + i, epoch_idx)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = one_step(first_step + args.warmup + i, epoch_idx)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = t.item()
        rec = {"rays_per_s": world * RAYS * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "final_loss": float(loss)}
        n_cam = int(trainer.n_samples.item())
        n_sun = int(trainer.out[:RAYS, 15].sum().item()) if wl == "full" else 0     # sc_pts_per_ray column (sat_rendering.py:311)
        rec["camera_samples_per_step"], rec["sun_samples_per_step"] = n_cam, n_sun
        # ---- untimed pass: the same steps with per-kernel HIP-event brackets ----
        kernels = {}
        if not args.no_kernel_pass:
            trainer.profile_enable(args.steps)
            for i in range(args.steps):
                one_step(first_step + args.warmup + args.steps + i, epoch_idx)
            torch.cuda.synchronize()
            prof = trainer.profile_read()
            trainer.profile_enable(0)
            dead = MAC_TRANSIENT if wl == "rgb" else 0          # transient head outside the graph when epoch_idx < 2 (s = 1, MSE on rgb)
            # bf16: the camera pass's trunk backward is layer-pipelined (csrc/eonerf_bwd_pipe.hip): ONE kernel does the dX chain AND the
            # 256 x 256 weight gradients of layers 1..7; the chain kernel keeps the heads, the GEMM the jobs the pipeline leaves
            piped = prof["bwd_pipe_camera"][1] > 0
            mac_trunk_dx, mac_trunk_dw = 7 * 65536, 7 * 65536
            flop_of = {"fwd_chain_camera": 2.0 * MAC_FWD * n_cam,
                       "bwd_chain_camera": 2.0 * (MAC_BWD - dead - (mac_trunk_dx if piped else 0)) * n_cam,
                       "bwd_pipe_camera": 2.0 * (mac_trunk_dx + mac_trunk_dw) * n_cam,
                       "wgrad_gemm": 2.0 * ((MAC_WGRAD - dead - (mac_trunk_dw if piped else 0)) * n_cam + MAC_DENS * n_sun),
                       "fwd_chain_sun": 2.0 * MAC_DENS * n_sun, "bwd_chain_sun": 2.0 * MAC_DENS * n_sun}
            # HBM bytes the design moves per launch (bf16 slabs): saved rows x element size (+ 32 B ReLU masks per slot)
            rows_w = 2496 if wl == "rgb" else 3012              # forward chain writes: enc 64 + X1..X8 2048 + bottleneck 256 + A1 128 (+ T 512 + emb 4)
            rows_g = 2180 if wl == "rgb" else 2694              # backward chain writes: dY0..7 2048 + dA1 128 + d sigma 1 + d albedo 3 (+ dT 512 + 2)
            rows_rd = (2564 + 2816) if wl == "rgb" else (3334 + 3972)     # weight-gradient GEMM reads both operands of every job once
            masks = 9 if wl == "rgb" else 13
            rows_pipe = 0
            if piped:
                # heads chain: writes dY_7 in unit order (256 rows) + the head gradient rows (A1 128, A2 32, sigma 32 [+ T 512 + 32]);
                # pipeline: reads dY_7 (256) and X1..X7 (7 x 256), writes dY_5 and dY_0 (2 x 256) for the GEMM jobs it leaves;
                # GEMM: layer 0 and the skip columns of layer 5 (2 x (256 + 64)), sigma (32 + 256), bottleneck factors and head jobs
                rows_g = 256 + 192 + (0 if wl == "rgb" else 544)
                rows_pipe = 256 + 7 * 256 + 2 * 256
                # (rows actually fetched: layer 0 and skip columns 2 x (256 + 64), sigma 1 + 256, a2 3 + 128; rgb: bottleneck factor and
                #  albedo layer 1 (128 + 256 each); full: dY A1 | dY T1 share one block, so [M_a; M_t] and the two first head layers are
                #  one 256 + 256 job each, + embedding columns 128 + 4, T2..T4 3 x 256, the two one-row heads 2 + 128)
                rows_rd = 2 * 320 + 257 + 131 + (2 * 384 if wl == "rgb" else 2 * 512 + 132 + 3 * 256 + 130)
            bytes_of = {"fwd_chain_camera": (rows_w * elt + masks * 32) * n_cam, "bwd_chain_camera": (rows_g * elt + masks * 32) * n_cam,
                        "bwd_pipe_camera": (rows_pipe * elt if piped else 0) * n_cam,
                        "wgrad_gemm": rows_rd * elt * n_cam + ((2 * 320 + 257) if piped else (2305 + 2176)) * elt * n_sun,
                        "fwd_chain_sun": (2112 * elt + 8 * 32) * n_sun, "bwd_chain_sun": ((2048 + 1) * elt + 8 * 32) * n_sun}
            for name, flop in flop_of.items():
                ms, cnt = prof[name]
                if cnt:
                    avg = ms / cnt
                    kernels[name] = {"avg_ms": avg, "tflops": flop / (avg * 1e-3) / 1e12, "frac_mfma": flop / (avg * 1e-3) / 1e12 / peak,
                                     "algorithmic_flop_per_launch": flop, "design_hbm_bytes_per_launch": bytes_of[name],
                                     "hbm_gbps": bytes_of[name] / (avg * 1e-3) / 1e9}
            rec["kernels"] = kernels
            rec["kernel_ms_sum"] = sum(k["avg_ms"] for k in kernels.values())
        # ---- whole-step MFMA fraction: useful FLOPs per step / step time / peak ----
        dead = MAC_TRANSIENT if wl == "rgb" else 0
        pruned = 2.0 * ((MAC_FWD + MAC_BWD - dead + MAC_WGRAD - dead) * n_cam + 3 * MAC_DENS * n_sun)
        s8d = 3.0 * (F_CAM * n_cam + F_DEN * n_sun)             # train = 3 x forward FLOPs at the measured sample counts
        s8d_nominal = 3.0 * 127 * RAYS * (F_CAM + (F_DEN if wl == "full" else 0))
        sec = rec["ms_per_step"] * 1e-3                         # per GPU: RAYS rays per step and rank (weak scaling)
        rec["step_mfma_frac"] = {"kernels_useful_flop": pruned / sec / 1e12 / peak,
                                 "survey_8d_3xF_measured_samples": s8d / sec / 1e12 / peak,
                                 "survey_8d_3xF_nominal_127": s8d_nominal / sec / 1e12 / peak}
        rec["step_tflops_per_gpu"] = s8d / sec / 1e12
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
            kd = kernels[dom]
            key = ("rgb_" if wl == "rgb" else "full_") + args.precision
            rec["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kd["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": kd["frac_mfma"], "frac_mfma": kd["frac_mfma"],
                               "algorithmic_flop_per_launch": kd["algorithmic_flop_per_launch"], "avg_launch_ms": kd["avg_ms"],
                               "traffic": TRAFFIC.get(key, {}).get(dom),
                               "hbm": {"achieved_gbps": kd["hbm_gbps"], "peak_gbps": PEAK_HBM_GBPS, "frac": kd["hbm_gbps"] / PEAK_HBM_GBPS,
                                       "design_bytes_per_launch": kd["design_hbm_bytes_per_launch"],
                                       "note": "bytes the design stashes/re-reads, not SURVEY 8(d)'s compulsory 148 B/ray"}}
        return rec

    workloads = ("rgb", "full") if args.workload == "both" else (args.workload,)
    recs, first = {}, 0
    for wl in workloads:
        recs[wl] = measure(wl, first)
        if rank == 0:
            print(f"[bench] {wl}: {recs[wl]['rays_per_s']:.0f} rays/s, {recs[wl]['ms_per_step']:.3f} ms/step", file=sys.stderr, flush=True)
        first += args.warmup + 2 * args.steps
    if rank == 0:
        head = recs[workloads[0]]
        names = {"rgb": "JAX_068-like synthetic rays, sigma+albedo path (shadow pass off, epoch<2, MSE), 4096 rays x 128 samples per GPU",
                 "full": "JAX_068-like synthetic rays, full EO-NeRF (shadow-ray pass + sun-visibility head + uncertainty loss), "
                         "4096 rays x 128 samples per GPU"}
        result = {
            "metric": "train rays/sec on JAX_068 (4096 rays x 128 samples)", "value": head["rays_per_s"], "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": names[workloads[0]], "rays_per_gpu": RAYS, "n_samples": 128, "n_images": N_IMG, "parallelism": f"dp{world}",
                       "batching": "GPU-resident ray table, shuffled on the device once per epoch, batches = slices (RayTable.batch); in-kernel Philox jitter",
                       "camera_samples_per_step": head["camera_samples_per_step"], "final_loss": head["final_loss"]},
            "roofline": head.get("roofline"),
            "kernels": head.get("kernels"),
            "step_mfma_frac": head["step_mfma_frac"],
        }
        if "full" in recs and workloads[0] != "full":
            f = recs["full"]
            result["full"] = {"workload": names["full"], "value": f["rays_per_s"], "unit": "rays/s", "ms_per_step": f["ms_per_step"],
                              "camera_samples_per_step": f["camera_samples_per_step"], "sun_samples_per_step": f["sun_samples_per_step"],
                              "roofline": f.get("roofline"), "kernels": f.get("kernels"), "step_mfma_frac": f["step_mfma_frac"],
                              "final_loss": f["final_loss"]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(workloads)
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
