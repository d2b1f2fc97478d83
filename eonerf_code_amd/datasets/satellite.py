"""Ray containers of the reference (datasets/satellite.py:21-30) -- the only part of the dataset module that sits on
the hot path.  Ray generation itself (RPC localisation) is SURVEY.md 8f row N2."""
import collections

SatRays = collections.namedtuple("Rays", ("origins", "viewdirs", "sundirs", "img_idx", "t_near", "t_far"))


def define_satrays_from_tensors(rays, ts):
    """rays [N,11] = (origin3, dir3, near1, far1, sun3), ts [N,1] int64 -> SatRays of column views (no copy)."""
    return SatRays(origins=rays[:, :3], viewdirs=rays[:, 3:6], sundirs=rays[:, 8:11], img_idx=ts,
                   t_near=rays[:, 6:7], t_far=rays[:, 7:8])


def namedtuple_map(fn, tup):
    """Apply fn to every field of a namedtuple (datasets/satellite.py:28-30)."""
    return type(tup)(*(None if x is None else fn(x) for x in tup))


def satrays_to_table(rays: "SatRays"):
    """Inverse of define_satrays_from_tensors: a contiguous fp32 [N,11] table + int64 [N] image indices.
    Zero-copy when the fields are still column views of one [N,11] tensor."""
    import torch
    o = rays.origins
    base = getattr(o, "_base", None)
    if (base is not None and base.dim() == 2 and base.shape[1] == 11 and base.is_contiguous() and base.dtype == torch.float32
            and o.data_ptr() == base.data_ptr() and rays.sundirs.data_ptr() == base.data_ptr() + 32
            and rays.viewdirs.data_ptr() == base.data_ptr() + 12 and base.shape[0] == o.shape[0]):
        table = base
    else:
        table = torch.cat([rays.origins, rays.viewdirs, rays.t_near, rays.t_far, rays.sundirs], dim=1).float().contiguous()
    img = rays.img_idx.reshape(-1).to(torch.int64).contiguous()
    return table, img
