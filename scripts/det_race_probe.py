"""Round-5 diagnostic: bit-reproducibility of the DETERMINISTIC backward, many repetitions on one forward.  The library named by EONERF_LIB
(default: the shipped one) runs one training forward of a full EO-NeRF step (shadow pass on) and then the backward N times into a zeroed
gradient buffer; every repetition must reproduce the first one bit for bit.  (ab_libs/libeonerf_oldreduce.so = the weight-gradient reduce of
round 4, whose camera / shadow jobs of one layer raced on a read-modify-write.)"""
import os, sys, torch
os.environ["EONERF_DETERMINISTIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eonerf_code_amd import _lib
from eonerf_code_amd.synthetic import synthetic_batch
from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
from eonerf_code_amd.trainer import FusedTrainer
R, N = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
torch.manual_seed(1)
f = EONerfMLP(19, radiometric_normalization=True, precision="bf16").cuda()
tr = FusedTrainer(f, lr=0.0, max_rays=R)
rays, img, pix = (t.cuda() for t in synthetic_batch(R, 19, seed=5))
flags = _lib.F_TRAIN | _lib.F_SHADOWS
tr._render_forward(rays, img, R, flags, (None, None, None))
ref, bad, names = None, 0, []
for it in range(N):
    tr._grad_clean = False                      # -> d_flat.zero_() in front of the backward
    tr._render_backward(rays, img, R, flags, pixels=pix, kind=1)
    g = tr.d_flat[:tr.n_params]
    if ref is None:
        ref = g.clone()
    elif not torch.equal(g, ref):
        bad += 1
        if len(names) < 5:
            d = (g - ref).abs()
            names.append([(n, int((d[o:o + r * c] > 0).sum())) for n, o, r, c in f._layout if (d[o:o + r * c] > 0).any()])
tr.check_device_status()
print(f"{os.path.basename(os.environ.get('EONERF_LIB', 'shipped'))}: {bad} of {N - 1} repetitions differ from the first (R = {R}); tensors: {names}")
