#!/usr/bin/env python3
"""Data-parallel EO-NeRF training launcher -- the loop of train_eonerf.py:96-161,304 on the HIP hot path.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m eonerf_code_amd.train_dp --rays table.pt --n_images 19 --batch_size 4096 --max_train_steps 300000

`--rays` is a torch file {"rays": [N,11] fp32 normalised rays, "ts": [N] int64 image index, "rgbs": [N,3]} as
datasets/satellite.py:406-481 builds it (eonerf_code_amd.datasets.satellite.generate_rays produces the rays on the
GPU); without it a synthetic JAX_068-like table is used.  Same schedule as the reference: seed 42, Adam lr 5e-4,
StepLR gamma 0.9 per epoch, MSE for epoch < 2 then the uncertainty loss with the shadow pass on.
"""
import argparse
import os
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", default=None)
    ap.add_argument("--n_images", type=int, default=19)
    ap.add_argument("--batch_size", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--max_train_steps", type=int, default=1000)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--n_samples", type=int, default=128, help="samples per ray = int(2 / render_step_size) (opt.py:54): 2 .. 256")
    ap.add_argument("--logs_dir", default="logs")
    ap.add_argument("--exp_name", default="eonerf_hip")
    ap.add_argument("--synthetic_rays", type=int, default=1 << 20)
    ap.add_argument("--check_every", type=int, default=1000, help="steps between host syncs (loss print + device status on every rank)")
    ap.add_argument("--dump_params", default=None, help="write the final flat parameters of every rank to <path>.rank<r>")
    args = ap.parse_args()

    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    # EONERF_DP_REHEARSAL=1 (tests on a one-GPU box, not a deployment mode): every rank on cuda:0, gloo instead of RCCL; run it with
    # EONERF_PIPE=0 -- the pipelined backward assumes the card to itself
    rehearsal = os.environ.get("EONERF_DP_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # (EONERF_FORCE_ALLREDUCE=1: the process group and the gradient exchange also at world size 1 -- the N > 1 code path on a one-GPU box)
    if world > 1 or os.environ.get("EONERF_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from .checkpoint import save_checkpoint
    from .radiance_fields.eonerf import EONerfMLP
    from .synthetic import synthetic_batch
    from .trainer import FusedTrainer, RayTable

    torch.manual_seed(42)                                                   # train_eonerf.py:37
    if args.rays:
        d = torch.load(args.rays, map_location="cpu")
        rays, ts, rgbs = d["rays"], d["ts"], d["rgbs"]
    else:
        rays, ts, rgbs = synthetic_batch(args.synthetic_rays, args.n_images)
    table = RayTable(rays, ts, rgbs, dev, seed=42, rank=rank, world=world)
    field = EONerfMLP(args.n_images, radiometric_normalization=True, precision=args.precision).to(dev)
    trainer = FusedTrainer(field, lr=args.lr, max_rays=args.batch_size, keep_message=False, n_samples=args.n_samples)
    trainer.set_noise_seed(42 + 1000003 * rank)                              # per-rank jitter stream
    steps_per_epoch = max(1, table.steps_per_epoch(args.batch_size))
    step, tic = 0, time.time()
    for epoch in range(10 ** 7):
        for i in range(steps_per_epoch):
            r, im, px = table.batch(epoch, i, args.batch_size)
            nxt = None
            if trainer._exchanges() and i + 1 < steps_per_epoch:                     # the next batch's sampler runs under this step's gradient exchange
                r2, im2, _ = table.batch(epoch, i + 1, args.batch_size)
                nxt = (r2, im2, epoch)
            loss = trainer.step(r, im, px, epoch, next_batch=nxt)
            if step % args.check_every == 0:                                # the only host sync, every 1000 steps (:173-178)
                # on EVERY rank: raises (-> non-zero exit of the job) if a device-side hand-off timed out on ANY rank since the last
                # check; the fault flag of the gradient message has kept all replicas from applying an update since
                trainer.check_device_status()
            if step % args.check_every == 0 and rank == 0:
                el = time.time() - tic
                print(f"epoch={epoch} | elapsed_time={el:.2f}s | step={step} | loss={float(loss):.5f} | "
                      f"rays/s={(step + 1) * args.batch_size * world / max(el, 1e-9):.0f}", flush=True)
            save_now = step > 0 and step % (4 * steps_per_epoch) == 0         # save_freq, :180-191 (the same decision on every rank)
            if save_now and step % args.check_every != 0:
                # a checkpoint must not hold updates that were skipped: after a device-side fault the Adam kernel leaves the weights alone
                # while the host's step count keeps running -- every rank checks (and raises together) BEFORE rank 0 writes
                trainer.check_device_status()
            if save_now and rank == 0:
                save_checkpoint(os.path.join(args.logs_dir, args.exp_name, f"ckpts/epoch={epoch}.ckpt"), epoch, field, trainer, loss)
            if step == args.max_train_steps:
                trainer.check_device_status()
                if args.dump_params:                                        # replica-equality checks of the tests
                    torch.save(field.flat_params().detach().cpu(), f"{args.dump_params}.rank{rank}")
                if torch.distributed.is_initialized():
                    torch.distributed.destroy_process_group()
                return
            step += 1
        trainer.set_lr(trainer.lr * 0.9)                                    # StepLR(step_size=1, gamma=0.9), :64,304


if __name__ == "__main__":
    main()
