"""One rank of the data-parallel GPU test (tests/test_dp_gpu.py): a fresh process that shares cuda:0 with its peer.
    python tests/dp_worker.py RANK WORLD PORT OUT_DIR MODE EPOCH
MODE "ok": one FusedTrainer.step on this rank's slice of a shared 2R batch (injected noise), then the reduced gradient and the
updated parameters are saved.  MODE "fail": rank 1 hands the C ABI an invalid argument: its non-zero return code must end the
job (non-zero exit), not hang it.
MODE "pipe": the bf16 path with the layer-pipelined backward.  That kernel assumes the card to itself, so the two ranks take TURNS
for the render/backward half of the step (barriers in between) and then meet in the all-reduce + Adam half.
MODE "pipedet": as "pipe" with fixed-order gradient sums (EONERF_DETERMINISTIC=1): two runs give bit-identical results, which is what the
two-bucket exchange is compared on (EONERF_EXCHANGE_BUCKETS=1 / 2).
MODE "fault": as "pipe", but rank 1's context was created with EONERF_PIPE_FAULT=3 (one stage never publishes its tiles): its
watchdog fires, the fault flag travels in the gradient message, NEITHER rank applies the update, and BOTH ranks raise.
MODE "nccl1": world size 1 over RCCL (backend "nccl") with EONERF_FORCE_ALLREDUCE=1: the pipelined step with the collective on the
side stream, in the process group the 8-GPU job uses.
MODE "pretog": as "pre0", with FusedTrainer.set_exchange_buckets switching between one and two buckets from step to step.
MODE "pre1" / "pre0": world size 1 over RCCL with the forced all-reduce, deterministic backward, production (Philox) noise: five steps over
a small ray table with / without the next batch's sampler enqueued under the exchange (FusedTrainer.step(next_batch=...)); the test compares
the two runs' parameters bit for bit.
Exit code 0 only on success."""
import datetime
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    rank, world, port, out_dir, mode, epoch = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    backend = "nccl" if mode in ("nccl1", "pre0", "pre1", "pretog") else "gloo"
    if mode in ("pre0", "pre1", "pretog", "pipedet"):
        os.environ["EONERF_DETERMINISTIC"] = "1"                      # read when the context is created
    if backend == "nccl":
        os.environ["EONERF_FORCE_ALLREDUCE"] = "1"
        torch.cuda.set_device(0)
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60),
                                             device_id=torch.device("cuda", 0))
    else:
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    torch.cuda.set_device(0)
    from oracle import eonerf_oracle as orc                       # test infrastructure: seeded weights / rays only
    from eonerf_code_amd import _lib
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP, _ptr, _stream
    from eonerf_code_amd.trainer import FusedTrainer, rank_slice
    n_img, R = 4, 256
    piped = mode in ("pipe", "pipedet", "fault", "nccl1", "pre0", "pre1", "pretog")
    # rank 1 starts from DIFFERENT weights: the trainer's initial broadcast must make the replicas identical
    sd = orc.random_state_dict(n_img, seed=91 + 7 * rank, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    f = EONerfMLP(n_img, radiometric_normalization=True, precision="bf16" if piped else "fp32")
    f.load_state_dict(sd, strict=True)
    f = f.cuda()
    if mode == "fault" and rank == 1:
        os.environ["EONERF_PIPE_FAULT"] = "3"                        # read when the context is created
    tr = FusedTrainer(f, lr=5e-4, max_rays=R)
    os.environ.pop("EONERF_PIPE_FAULT", None)
    assert tr.world == world
    rays, ts, rgbs, u_cam, u_sun = orc.synthetic_batch(R * world, n_img, seed=92)
    a, b = rank_slice(R * world, rank, world)
    dev = torch.device("cuda", 0)
    sl = lambda t: t[a:b].contiguous().to(dev)
    if mode == "fail" and rank == 1:
        L = _lib.lib()
        _lib.check(L.eonerf_render_forward(tr.ctx, None, None, None, None, None, None, None, R, _lib.F_TRAIN, None, None, None, 0, _stream()))
        raise SystemExit("unreachable: the C ABI accepted null pointers")
    if mode in ("pre0", "pre1", "pretog"):
        from eonerf_code_amd.trainer import RayTable
        tr.set_noise_seed(5)
        big = orc.synthetic_batch(R * 5, n_img, seed=93)
        table = RayTable(big[0], big[1], big[2], dev, seed=7)
        calls = []
        orig = tr._presample
        tr._presample = lambda *a: (calls.append(1), orig(*a))[1]
        losses = []
        for i in range(5):
            if mode == "pretog":                                   # the exchange's order switched between steps (bench.py's untimed A/B does that)
                tr.set_exchange_buckets(1 + i % 2)
            r, im, px = table.batch(0, i, R)
            nxt = (table.batch(0, i + 1, R)[0], table.batch(0, i + 1, R)[1], epoch) if (mode == "pre1" and i < 4) else None
            losses.append(tr.step(r, im, px, epoch, next_batch=nxt))
        torch.cuda.synchronize()
        tr.check_device_status()
        assert len(calls) == (4 if mode == "pre1" else 0), calls
        torch.save({"flat": tr.flat.detach().cpu(), "loss": [float(l) for l in losses], "n_samples": int(tr.n_samples.item())},
                   os.path.join(out_dir, f"{mode}.pt"))
        torch.distributed.destroy_process_group()
        return
    args = (sl(rays), sl(ts.reshape(-1)), sl(rgbs), epoch)
    noise = (sl(u_cam), None, sl(u_sun))
    p_before = tr.flat.detach().cpu().clone()
    if piped:
        loss = None
        for r in range(world):                                     # one rank at a time on the card
            if r == rank:
                loss = tr.forward_backward(*args, noise=noise)
                torch.cuda.synchronize()
            torch.distributed.barrier()
        tr.reduce_and_update()
    else:
        loss = tr.step(*args, noise=noise)
    torch.cuda.synchronize()
    torch.save({"d_flat": tr.d_flat.cpu(), "flat": tr.flat.detach().cpu(), "flat_before": p_before, "loss": float(loss)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    try:
        tr.check_device_status()                                    # every rank: raises when ANY rank's gradients were invalid
    finally:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
