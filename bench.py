#!/usr/bin/env python3
"""bench.py -- train rays/sec of the EO-NeRF hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete optimisation step on one synthetic ray batch per GPU that is already resident in HBM:
jitter noise -> render forward (sampler, fused MLP chain, compositing, shading) -> loss -> backward (compositing,
backward chain, weight-gradient GEMM) -> [RCCL all-reduce of the flat gradient] -> Adam + weight re-pack.
Workload at every N: BASELINE.json configs[1] -- JAX_068-like synthetic rays, 4096 rays x 128
samples per GPU, shadow pass off (epoch_idx < 2, MSE loss), bf16 MFMA; `--workload full` runs configs[2]
(shadow-ray pass + uncertainty loss).  Weak scaling: per-GPU work is fixed, rays are independent units.

Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (HIP-event timed inside the timed region through
the library's measurement hooks), `cpu_baseline` is the oracle (CPU port of the reference algorithm) timed on this
box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_IMG = 19            # JAX_068-like
RAYS = 4096
STEP_SIZE = 2.0 / 128
MAC_FWD = 672640      # MACs/sample inside the forward chain kernel: trunk 491,008 + sigma 256 + bottleneck 65,536
                      #   + albedo 33,152 + transient 82,688  (SURVEY.md 8a H6 minus the per-ray ambient head)
MAC_BWD = MAC_FWD - 63 * 256          # dX chain: no input gradient on the camera pass
MAC_WGRAD = MAC_FWD                   # one MAC per weight per sample
MAC_TRANSIENT = 82688                 # transient head: outside the autograd graph when epoch_idx < 2 (s = 1, MSE on rgb)
MAC_DENS_FWD = 491008 + 256
MAC_DENS_BWD = 491008 + 256           # incl. input gradient through layer 0 / skip columns
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3
# HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB units; see
# profiles/README.md) on the default workload; null for configurations that were not profiled
TRAFFIC = {}
try:
    with open(os.path.join(REPO, "profiles", "traffic.json")) as _f:
        TRAFFIC = json.load(_f).get("rgb_bf16", {})
except OSError:
    pass


def cpu_baseline(workload, n_rays=128, reps=1):
    """The oracle (CPU restatement of the reference algorithm, torch fp32) timed on the host cores."""
    from oracle import eonerf_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 64))      # more threads only add contention at this batch size
    epoch = 3 if workload == "full" else 0
    sd = orc.random_state_dict(N_IMG, seed=42)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    opt = torch.optim.Adam([v for v in params.values() if v.is_floating_point()], lr=5e-4)
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(n_rays, N_IMG, seed=1234)
    orc.train_step(params, rays, ts, rgbs, u_cam, u_sun, epoch, STEP_SIZE, opt)        # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.train_step(params, rays, ts, rgbs, u_cam, u_sun, epoch, STEP_SIZE, opt)
    dt = (time.perf_counter() - t0) / reps
    return {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} full train steps (render+loss+backward+Adam) of {n_rays} rays x 128 samples, torch CPU fp32, "
                      f"{dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("rgb", "full"), default="rgb")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from eonerf_code_amd.synthetic import synthetic_batch
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    from eonerf_code_amd.trainer import FusedTrainer

    torch.manual_seed(42)                                     # train_eonerf.py:37
    field = EONerfMLP(N_IMG, radiometric_normalization=True, precision=args.precision).to(dev)
    trainer = FusedTrainer(field, lr=5e-4, max_rays=RAYS)
    epoch = 3 if args.workload == "full" else 0

    # a table of synthetic rays resident in HBM; every step takes a different contiguous batch of it
    n_batches = 8
    rays, img, rgbs = (t.to(dev) for t in synthetic_batch(RAYS * n_batches, N_IMG, seed=1234 + rank))
    torch.manual_seed(1000 + rank)

    def one_step(i):
        b = (i % n_batches) * RAYS
        return trainer.step(rays[b:b + RAYS], img[b:b + RAYS], rgbs[b:b + RAYS], epoch)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    trainer.profile_enable(args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()

    prof = trainer.profile_read()
    n_cam = int(trainer.n_samples.item())
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        kernels = {}
        dead = MAC_TRANSIENT if args.workload == "rgb" else 0          # work the backward kernels really skip
        mac_of = {"fwd_chain_camera": MAC_FWD, "bwd_chain_camera": MAC_BWD - dead, "wgrad_gemm": MAC_WGRAD - dead}
        for name, macs in mac_of.items():
            ms, cnt = prof[name]
            if cnt:
                kernels[name] = {"avg_ms": ms / cnt, "tflops": 2.0 * macs * n_cam / (ms / cnt * 1e-3) / 1e12}
        if args.workload == "full":      # shadow-pass chains: launch time only (their sample count differs from the camera pass)
            for name in ("fwd_chain_sun", "bwd_chain_sun"):
                ms, cnt = prof[name]
                if cnt:
                    kernels[name] = {"avg_ms": ms / cnt}
        dom = max(mac_of, key=lambda k: kernels[k]["avg_ms"] if k in kernels else 0.0)
        macs = mac_of[dom]
        step_flops = 2.0 * sum(mac_of.values()) * n_cam
        # HBM bytes each kernel must move per sample (bf16 slabs): rows x 2 B
        #   forward chain writes   enc 64 + X1..X8 2048 + bottleneck 256 + albedo hidden 128 (+ transient 512 + emb 4) rows, 9 (13) masks
        #   backward chain writes  dY0..7 2048 + dA1 128 + d sigma 1 + d albedo 3 (+ dT1..4 512 + d ts/tb 2) rows, reads the masks
        #   weight-gradient GEMM   reads every job's two operands once (valid rows; camera-pass jobs only for --workload full)
        rows_rd_wgrad = (2564 + 2816) if args.workload == "rgb" else (3334 + 3972)
        bytes_of = {"fwd_chain_camera": (2496 if args.workload == "rgb" else 3012) * 2 + (9 if args.workload == "rgb" else 13) * 32,
                    "bwd_chain_camera": (2180 if args.workload == "rgb" else 2694) * 2 + (9 if args.workload == "rgb" else 13) * 32,
                    "wgrad_gemm": rows_rd_wgrad * 2}
        elt = 2 if args.precision == "bf16" else 4
        for name in bytes_of:
            if name in kernels:
                kernels[name]["hbm_gbps"] = bytes_of[name] * (elt / 2) * n_cam / (kernels[name]["avg_ms"] * 1e-3) / 1e9
        if dom == "wgrad_gemm":      # 131 FLOP/B: HBM-bound
            roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["hbm_gbps"], "peak": 8000.0, "unit": "GB/s",
                        "frac": kernels[dom]["hbm_gbps"] / 8000.0, "traffic": TRAFFIC.get(dom),
                        "algorithmic_bytes_per_launch": bytes_of[dom] * (elt / 2) * n_cam, "avg_launch_ms": kernels[dom]["avg_ms"],
                        "mfma_tflops": kernels[dom]["tflops"]}
        else:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": kernels[dom]["tflops"] / peak, "traffic": TRAFFIC.get(dom),
                        "algorithmic_flop_per_launch": 2.0 * macs * n_cam, "avg_launch_ms": kernels[dom]["avg_ms"]}
        result = {
            "metric": "train rays/sec on JAX_068 (4096 rays x 128 samples)", "value": world * RAYS * args.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("JAX_068-like synthetic rays, sigma+albedo path (shadow pass off, epoch<2, MSE), 4096 rays x 128 samples per GPU"
                                    if args.workload == "rgb" else
                                    "JAX_068-like synthetic rays, full EO-NeRF (shadow-ray pass + uncertainty loss), 4096 rays x 128 samples per GPU"),
                       "rays_per_gpu": RAYS, "n_samples": 128, "n_images": N_IMG, "parallelism": f"dp{world}",
                       "camera_samples_per_step": n_cam, "final_loss": float(loss)},
            "roofline": roofline,
            "kernels": kernels,
            "step_mfma_frac": step_flops / (ms_step * 1e-3) / 1e12 / peak,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
