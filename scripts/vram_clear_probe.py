"""is the slow start of a fresh process the driver's deferred clearing of freshly allocated VRAM?
Warm the trainer up completely, then hipMalloc a NEW block of G GB (torch.empty of a size the caching allocator cannot serve from its
pool) and time the next steps in blocks of 5 (HIP events) WITHOUT touching that block.  If the steps right behind the allocation are slow,
the slow start follows the allocation (background clear competing for HBM), not the kernels."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.trainer import FusedTrainer, RayTable
dev = torch.device("cuda", 0)
torch.manual_seed(42)
field = EONerfMLP(19, radiometric_normalization=True, precision="bf16").to(dev)
tr = FusedTrainer(field, lr=5e-4, max_rays=4096, keep_message=False)
table = RayTable(*synthetic_batch(4096 * 64, 19, seed=1234), dev, seed=42)
def steps(n, i0):
    for i in range(n):
        r, im, px = table.batch(0, (i0 + i) % 64, 4096)
        tr.step(r, im, px, 3)
def timed_blocks(nb, tag):
    out = []
    for b in range(nb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); steps(5, 5 * b); e1.record(); e1.synchronize()
        out.append(round(e0.elapsed_time(e1) / 5, 3))
    print(tag, out, flush=True)
timed_blocks(8, "fresh process, ms/step in blocks of 5:")
timed_blocks(4, "warm:")
for gb in (4, 16, 48):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(gb << 30, dtype=torch.uint8, device=dev)          # a new hipMalloc (nothing that large in the pool)
    t_alloc = (time.perf_counter() - t0) * 1e3
    timed_blocks(6, f"right behind a fresh {gb}-GB allocation (host {t_alloc:.1f} ms):")
    del x
    torch.cuda.empty_cache()
    timed_blocks(3, f"after freeing it:")
