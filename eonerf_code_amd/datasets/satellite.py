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


# ---------------------------------------------------------------------------------------------------------------------
# RPC ray generation on the GPU: datasets/satellite.py:65-139,456-458 of the reference.
# `rpc` is the "rpc" dict of a dataset JSON (rpcm dict format) or any object with the same attribute names.
# ---------------------------------------------------------------------------------------------------------------------
_RPC_SCALARS = ("row_offset", "col_offset", "lat_offset", "lon_offset", "alt_offset",
                "row_scale", "col_scale", "lat_scale", "lon_scale", "alt_scale")


def _rpc_struct(rpc, downscale=1.0):
    from .. import _lib
    get = (lambda k: rpc[k]) if isinstance(rpc, dict) else (lambda k: getattr(rpc, k))
    s = _lib.EonerfRpc()
    for k in ("col_num", "col_den", "row_num", "row_den"):
        v = [float(x) for x in get(k)]
        if len(v) != 20:
            raise ValueError(f"rpc[{k}] must have 20 coefficients")
        setattr(s, k, (type(getattr(s, k)))(*v))
    for k in _RPC_SCALARS:
        setattr(s, k, float(get(k)))
    if downscale != 1.0:        # sat_utils.rescale_rpc(rpc, 1/img_downscale), sat_utils.py:41-59
        alpha = 1.0 / float(downscale)
        for k in ("row_scale", "col_scale", "row_offset", "col_offset"):
            setattr(s, k, getattr(s, k) * alpha)
    return s


def utm_zone_from_lonlat(lon, lat):
    """(zone number, south?) as utm.latlon_to_zone_number / latitude_to_zone_letter give them (sat_utils.py:105-109)."""
    if 56 <= lat < 64 and 3 <= lon < 12:
        zone = 32
    elif 72 <= lat <= 84 and lon >= 0 and lon < 42:
        zone = 31 if lon < 9 else 33 if lon < 21 else 35 if lon < 33 else 37
    else:
        zone = int((lon + 180) / 6) % 60 + 1
    return zone, lat < 0


def generate_rays(rpc, min_alt, max_alt, h=None, w=None, cols=None, rows=None, img_downscale=1.0, sun_elevation_deg=None,
                  sun_azimuth_deg=None, scene_offset=None, scene_scale=None, zone=None, south=None, device="cuda", want_raw=False):
    """Rays of one image.  With scene_offset/scale (scene.loc_utm X/Y/Z) -> normalised fp32 [N,11] rays (what load_data keeps,
    datasets/satellite.py:456-478); want_raw additionally (or alone) returns the un-normalised [N,8] rays of get_rays
    (:65-121), the payload of the reference's cache files.  Pixels: the full h x w grid, or explicit cols/rows."""
    import ctypes as C
    import torch
    from .. import _lib
    from ..radiance_fields.eonerf import _ptr, _stream
    s = _rpc_struct(rpc, img_downscale)
    if zone is None:
        zone, south = utm_zone_from_lonlat(s.lon_offset, s.lat_offset)
    dev = torch.device(device)
    if cols is not None:
        c = torch.as_tensor(cols, dtype=torch.float64).reshape(-1).to(dev).contiguous()
        r = torch.as_tensor(rows, dtype=torch.float64).reshape(-1).to(dev).contiguous()
        n, width = c.numel(), 1
    else:
        c = r = None
        n, width = int(h) * int(w), int(w)
    normalise = scene_offset is not None
    rays = torch.empty(n, 11, dtype=torch.float32, device=dev) if normalise else None
    raw = torch.empty(n, 8, dtype=torch.float32, device=dev) if (want_raw or not normalise) else None
    off = (C.c_float * 3)(*[float(x) for x in scene_offset]) if normalise else None
    sc = (C.c_float * 3)(*[float(x) for x in scene_scale]) if normalise else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().eonerf_generate_rays(C.byref(s), _ptr(c), _ptr(r), n, width, float(min_alt), float(max_alt), int(zone),
                                                   1 if south else 0, float(sun_elevation_deg or 0.0), float(sun_azimuth_deg or 0.0),
                                                   off, sc, _ptr(raw), _ptr(rays), _stream()))
    if normalise and want_raw:
        return rays, raw
    return rays if normalise else raw


def get_rays(cols, rows, rpc, min_alt, max_alt, utm=True, device="cuda"):
    """datasets/satellite.py:65-121 (utm branch): float32 [N,8] = origin3, dir3, near, far -- on the GPU."""
    if not utm:
        raise NotImplementedError("the ECEF branch (--ecef) is not part of the hot path")
    return generate_rays(rpc, min_alt, max_alt, cols=cols, rows=rows, device=device)
