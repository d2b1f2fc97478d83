"""How the DSM MAE and its run-to-run spread evolve with training length (bf16 mode only: 4 ms per step) -- sizing of the twin-training
experiment (scripts/twin_training.py).  Usage: python scripts/twin_converge.py 2000 6000 12000"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from bf16_common import twin_train, export_quality  # noqa: E402
kw = {}
if os.environ.get("TWIN_DENSE") == "1":      # depth prior on every ray, constant weight (tests/bf16_common.py::train_on_terrain's supervision)
    kw = dict(prior_every=1, w_depth0=10.0, w_decay=1.0)
for steps in [int(a) for a in sys.argv[1:]]:
    res = []
    for seed in (7, 8, 9):
        t0 = time.time()
        q = export_quality(twin_train("bf16", steps=steps, steps_per_epoch=steps // 4, noise_seed=seed, **kw))
        res.append(q)
        print(f"steps {steps} jitter {seed}: DSM MAE {100 * q['dsm_mae_m']:.2f} cm PSNR {q['psnr']:.2f} ({time.time() - t0:.0f} s)", flush=True)
    m = [100 * r["dsm_mae_m"] for r in res]
    print(f"steps {steps}: MAE mean {sum(m) / 3:.2f} cm, max-min {max(m) - min(m):.2f} cm", flush=True)
