"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Restatement of the three nerfacc v0.5.2 functions the reference's hot path calls.

nerfacc is a third-party dependency pinned by the reference at git tag v0.5.2
(/root/reference/setup_env.sh:10) and is NOT vendored under /root/reference, so
its arithmetic cannot be executed here.  The functions below restate the
published algorithm of `nerfacc/volrend.py` @ v0.5.2 from its public API
documentation:

    render_transmittance_from_density:
        sigmas_dt = sigmas * (t_ends - t_starts)
        alphas    = 1 - exp(-sigmas_dt)
        trans     = exp(-exclusive_sum(sigmas_dt, per ray))
    render_weight_from_density:
        weights   = trans * alphas
    accumulate_along_rays:
        out[n_rays, C].index_add_(0, ray_indices, weights[:, None] * values)

PARITY STATUS: **unpinned** -- the reference ships no test or fixture at this
boundary.  Anchors: the reference call sites
(radiance_fields/eonerf.py:186-193,229-242; sat_rendering.py:106-110) and the
in-tree dense formulation `weights_from_sigma` (radiance_fields/eonerf.py:37-54)
which tests/test_oracle_golden.py cross-checks against.
"""
import torch


def _segment_starts(ray_indices: torch.Tensor, n_rays: int):
    """counts/offsets of each ray's contiguous run in a sorted ray_indices vector."""
    counts = torch.bincount(ray_indices, minlength=n_rays)
    offsets = torch.cumsum(counts, 0) - counts
    return counts, offsets


def exclusive_sum(x: torch.Tensor, ray_indices: torch.Tensor, n_rays: int) -> torch.Tensor:
    """Per-ray exclusive prefix sum of a flattened, ray-sorted vector (fp32, left to right)."""
    if x.numel() == 0:
        return x.clone()
    counts, offsets = _segment_starts(ray_indices, n_rays)
    max_c = int(counts.max().item())
    pos = torch.arange(x.numel(), device=x.device) - offsets[ray_indices]
    dense = torch.zeros(n_rays, max_c, dtype=x.dtype, device=x.device)
    dense = dense.index_put((ray_indices, pos), x)
    incl = torch.cumsum(dense, dim=1)
    # shifted inclusive sum == running sum of the previous elements (no re-rounding)
    excl = torch.cat([torch.zeros_like(incl[:, :1]), incl[:, :-1]], dim=1)
    return excl[ray_indices, pos]


def render_transmittance_from_density(t_starts, t_ends, sigmas, ray_indices=None, n_rays=None):
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, ray_indices, n_rays))
    return trans, alphas


def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=None, n_rays=None):
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, ray_indices, n_rays)
    weights = trans * alphas
    return weights, trans, alphas


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    if values is None:
        src = weights[:, None]
    else:
        src = weights[:, None] * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype, device=src.device)
    out = out.index_add(0, ray_indices, src)
    return out
