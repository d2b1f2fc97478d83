"""A/B of FusedTrainer.step(next_batch=...) (eonerf_presample: the next step's camera sampler under the gradient exchange) on ONE GPU:
RCCL process group of world size 1 with EONERF_FORCE_ALLREDUCE=1, bench.py's workload (4096 rays x 128 samples, full EO-NeRF step),
alternating blocks of 100 steps with and without the hint.  What a one-GPU box can show: the ordering costs nothing and the step tail
(exchange + Adam + re-pack, event to event) -- the world-1 collective itself is close to empty, so the gain of hiding a real exchange
(min(exchange, ~20 us) per step) only shows at N > 1 (`dist.step_tail_us` of the driver's scaling run).
    python3 scripts/presample_ab.py [rounds]"""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["EONERF_FORCE_ALLREDUCE"] = "1"
import torch

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.trainer import FusedTrainer, RayTable

RAYS, N_IMG, BATCHES = 4096, 19, 16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(42)
field = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16").to(dev)
tr = FusedTrainer(field, lr=5e-4, max_rays=RAYS, keep_message=False)
table = RayTable(*synthetic_batch(RAYS * BATCHES, N_IMG, seed=1234), dev, seed=42)


def block(hint, n, first):
    tr.tail_events = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(n):
        i = (first + k) % BATCHES
        r, im, px = table.batch(0, i, RAYS)
        nxt = None
        if hint and i + 1 < BATCHES:
            r2, im2, _ = table.batch(0, i + 1, RAYS)
            nxt = (r2, im2, 3)
        tr.step(r, im, px, 3, next_batch=nxt)
    e1.record()
    torch.cuda.synchronize()
    tail = statistics.median(a.elapsed_time(b) * 1e3 for a, b in tr.tail_events)
    tr.tail_events = None
    return e0.elapsed_time(e1) / n, tail


block(False, 150, 0)          # conditioning
for r in range(rounds):
    for mode in ("side", "async"):
        tr.exchange_async = mode == "async"
        for hint in (False, True):
            ms, tail = block(hint, 100, 0)
            print(f"round {r} exchange={mode:5s} presample={'on ' if hint else 'off'}  {ms:.4f} ms/step   step tail (exchange + Adam + re-pack) {tail:.1f} us", flush=True)
tr.check_device_status()
torch.distributed.destroy_process_group()
