"""Measure what bf16 arithmetic does at the bench size (4096 rays x 128 samples): bf16-HIP vs fp32-HIP (the latter is pinned to
the oracle at 1e-4) on identical rays, noise and weights -- rgb, depth, altitude (Z_scale 50 m) and per-tensor gradient cosine,
for (a) Xavier-init weights and (b) a field trained for a few hundred steps on a synthetic terrain (depth + colour supervision,
so that sigma is concentrated around a surface).  Prints one JSON line per state; tests/test_bf16_fullsize.py asserts the bounds."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.bf16_common import compare_precisions, make_fields, terrain_batch, train_on_terrain  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    f16, f32 = make_fields(seed=42)
    print(json.dumps({"state": "xavier", **compare_precisions(f16, f32, seed=1)}))
    for k in (steps // 4, steps // 2, steps):
        f16, f32 = make_fields(seed=42)
        train_on_terrain(f16, k)
        f32.load_state_dict(f16.state_dict())
        print(json.dumps({"state": f"trained_{k}", **compare_precisions(f16, f32, seed=1)}))
