"""Synthetic JAX_068-like ray batches for benchmarks and the launcher's dry runs: origins on the top
face of the normalised cube, near-nadir view directions, one sun direction per image from (elevation, azimuth) as in
datasets/satellite.py:57-63,486-500, uniform target colours."""
import math

import torch


def dir_vec_from_el_az(elevation_deg, azimuth_deg):
    el, az = math.radians(90 - elevation_deg), math.radians(azimuth_deg)
    return [-math.sin(az) * math.cos(el), -math.cos(az) * math.cos(el), -math.sin(el)]


def synthetic_batch(n_rays, n_img, seed=1234):
    """-> rays [R,11] fp32 (o3 d3 near far sun3), img_idx [R] int64, rgbs [R,3]."""
    g = torch.Generator().manual_seed(seed)
    o = torch.empty(n_rays, 3)
    o[:, :2] = torch.rand(n_rays, 2, generator=g) * 1.8 - 0.9
    o[:, 2] = 0.98
    d = torch.cat([0.15 * torch.randn(n_rays, 2, generator=g), -torch.ones(n_rays, 1)], 1)
    d = d / d.norm(dim=1, keepdim=True)
    el = torch.rand(n_img, generator=g) * 40 + 30
    az = torch.rand(n_img, generator=g) * 180 + 90
    sun = torch.tensor([dir_vec_from_el_az(90 - float(e), float(a)) for e, a in zip(el, az)], dtype=torch.float32)
    sun = sun / sun.norm(dim=1, keepdim=True)
    img = torch.randint(0, n_img, (n_rays,), generator=g)
    rays = torch.cat([o, d, torch.zeros(n_rays, 1), 2 * torch.ones(n_rays, 1), sun[img]], 1)
    return rays, img, torch.rand(n_rays, 3, generator=g)
