"""A/B of the gradient exchange's bucketing on ONE GPU: RCCL process group of world size 1 with EONERF_FORCE_ALLREDUCE=1, bench.py's
workload (4096 rays x 128 samples, full EO-NeRF step), alternating blocks of 100 steps:
    no exchange | one bucket (all-reduce behind the backward's last kernel) | two buckets, 0 CUs kept free | two buckets, 8 CUs kept free
What a one-GPU box can show: what the machinery costs (the event the library records behind the camera pass' pipelined launch, the second
collective, the CUs the GEMM launch gives up) and where the early collective sits relative to the end of the backward.  What hiding a REAL
exchange gains only shows at N > 1 (`dist.exchange_hidden_us` of the driver's scaling run).
    python3 scripts/buckets_ab.py [rounds]"""
import ctypes as C
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
os.environ["EONERF_FORCE_ALLREDUCE"] = "1"
import torch

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from eonerf_code_amd import _lib
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.trainer import FusedTrainer, RayTable

RAYS, N_IMG, BATCHES = 4096, 19, 16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
table = RayTable(*synthetic_batch(RAYS * BATCHES, N_IMG, seed=1234), dev, seed=42)


def trainer(mode):
    os.environ["EONERF_FORCE_ALLREDUCE"] = "0" if mode == "none" else "1"
    os.environ["EONERF_EXCHANGE_BUCKETS"] = "1" if mode in ("none", "one") else "2"
    os.environ["EONERF_COMM_CUS"] = "0" if mode == "two/0" else "8"
    torch.manual_seed(42)
    field = EONerfMLP(N_IMG, radiometric_normalization=True, precision="bf16").to(dev)
    return FusedTrainer(field, lr=5e-4, max_rays=RAYS, keep_message=False)


MODES = ("none", "one", "two/0", "two/8")
trs = {}
for m in MODES:
    trs[m] = trainer(m)


def block(mode, n, marks=False):
    tr = trs[mode]
    os.environ["EONERF_FORCE_ALLREDUCE"] = "0" if mode == "none" else "1"      # (read per step by the trainer)
    tr.tail_events = [] if marks else None      # (the event marks cost queue time: step time and marks come from separate blocks)
    tr.hidden_events = [] if marks and mode.startswith("two") else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(n):
        r, im, px = table.batch(0, k % BATCHES, RAYS)
        tr.step(r, im, px, 3)
    e1.record()
    torch.cuda.synchronize()
    tail = statistics.median(a.elapsed_time(b) * 1e3 for a, b in tr.tail_events) if tr.tail_events else float("nan")
    early = hidden = float("nan")
    if tr.hidden_events:
        early = statistics.median(a.elapsed_time(b) * 1e3 for a, b, _ in tr.hidden_events)
        hidden = statistics.median(max(0.0, min(a.elapsed_time(b), a.elapsed_time(c))) * 1e3 for a, b, c in tr.hidden_events)
    tr.tail_events = None
    tr.hidden_events = None
    return e0.elapsed_time(e1) / n, tail, early, hidden


for m in MODES:
    block(m, 60)          # conditioning
for r in range(rounds):
    for m in MODES:
        ms = block(m, 100)[0]
        _, tail, early, hidden = block(m, 30, marks=True)
        print(f"round {r} exchange={m:6s} {ms:.4f} ms/step   step tail {tail:6.1f} us   early collective {early:6.1f} us, of it before the backward's end {hidden:6.1f} us", flush=True)
for m in MODES:
    trs[m].check_device_status()
