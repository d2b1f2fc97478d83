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
                  sun_azimuth_deg=None, scene_offset=None, scene_scale=None, zone=None, south=None, device="cuda", want_raw=False,
                  want_geo=False):
    """Rays of one image.  With scene_offset/scale (scene.loc_utm X/Y/Z) -> normalised fp32 [N,11] rays (what load_data keeps,
    datasets/satellite.py:456-478); want_raw additionally (or alone) returns the un-normalised [N,8] rays of get_rays
    (:65-121), the payload of the reference's cache files.  Pixels: the full h x w grid, or explicit cols/rows.
    want_geo: returns ONLY the fp64 intermediates [N,8] (lon, lat, east, north at max_alt, then at min_alt) -- the values of
    rpc.localization / utm_from_latlon before the fp32 cast of :119-120 (tests)."""
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
    geo = torch.empty(n, 8, dtype=torch.float64, device=dev) if want_geo else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().eonerf_generate_rays(C.byref(s), _ptr(c), _ptr(r), n, width, float(min_alt), float(max_alt), int(zone),
                                                   1 if south else 0, float(sun_elevation_deg or 0.0), float(sun_azimuth_deg or 0.0),
                                                   off, sc, _ptr(raw), _ptr(rays), _ptr(geo), _stream()))
    if want_geo:
        return geo
    if normalise and want_raw:
        return rays, raw
    return rays if normalise else raw


def get_rays(cols, rows, rpc, min_alt, max_alt, utm=True, device="cuda"):
    """datasets/satellite.py:65-121 (utm branch): float32 [N,8] = origin3, dir3, near, far -- on the GPU."""
    if not utm:
        raise NotImplementedError("the ECEF branch (--ecef) is not part of the hot path")
    return generate_rays(rpc, min_alt, max_alt, cols=cols, rows=rows, device=device)


# ---------------------------------------------------------------------------------------------------------------------
# Dataset side of ray generation: the metadata JSONs, the ray cache and scene.loc_utm of the reference
# (datasets/satellite.py:299-307, 374-404, 406-481), without the image pixels (rasterio is not part of this path).
#   <root>/<img>.json      keys img, height, width, rpc (rpcm dict), min_alt, max_alt, sun_elevation, sun_azimuth
#   <cache_dir>/<id>.data  torch.save of the UN-normalised fp32 rays [h*w, 8] of get_rays (or [h*w, 11] incl. sun dirs)
#   <root>/scene.loc_utm   JSON {X,Y,Z}_{scale,offset}
# ---------------------------------------------------------------------------------------------------------------------
def sun_direction(sun_elevation_deg, sun_azimuth_deg):
    """get_sun_dirs(90 - sun_elevation, sun_azimuth) of load_data (:456-457) -> get_dir_vec_from_el_az (:57-63): fp64 3-vector."""
    import math
    el, az = math.radians(float(sun_elevation_deg)), math.radians(float(sun_azimuth_deg))
    return [-math.sin(az) * math.cos(el), -math.cos(az) * math.cos(el), -math.sin(el)]


def normalize_rays(rays, scene_offset, scene_scale):
    """datasets/satellite.py:124-139 on the tensor's device: rays [N,8|11] holding the fp32 values of the cache (+ fp64 sun
    directions) -> float64 [N,8|11] normalised rays, arithmetic in fp64 as numpy does after the hstack with the fp64 sun
    directions (:458); scene_offset / scene_scale are the dataset's fp32 values (:303-307)."""
    import torch
    r = rays.to(torch.float64)
    off = torch.as_tensor(scene_offset, dtype=torch.float32).to(r.device, torch.float64)
    sc = torch.as_tensor(scene_scale, dtype=torch.float32).to(r.device, torch.float64)
    o = r[:, :3]
    e = o + r[:, 3:6] * r[:, 7:8]
    o_n, e_n = (o - off) / sc, (e - off) / sc
    d = e_n - o_n
    far = torch.linalg.norm(d, dim=1, keepdim=True)
    cols = [o_n, d / far, torch.zeros_like(far), far]
    if r.shape[1] == 11:
        s = r[:, 8:11] / sc
        cols.append(s / torch.linalg.norm(s, dim=1, keepdim=True))
    return torch.cat(cols, dim=1)


def rpc_scaling_params(v):
    """sat_utils.rpc_scaling_params (sat_utils.py:32-39): (scale, offset) of a vector."""
    lo, hi = float(v.min()), float(v.max())
    scale = (hi - lo) / 2
    return scale, lo + scale


def scene_loc_from_rays(raw_rays):
    """init_scaling_params (datasets/satellite.py:395-403): the scene.loc_utm dict from all un-normalised rays [N,>=8]."""
    import torch
    near = raw_rays[:, :3].double()
    far = near + raw_rays[:, 7:8].double() * raw_rays[:, 3:6].double()
    pts = torch.cat([near, far], 0)
    d = {}
    for k, name in enumerate("XYZ"):
        d[name + "_scale"], d[name + "_offset"] = rpc_scaling_params(pts[:, k])
    return d


def read_scene_loc(path):
    """-> (scene_offset[3], scene_scale[3]) as the dataset keeps them (:299-307)."""
    import json
    with open(path) as f:
        d = json.load(f)
    return ([float(d["X_offset"]), float(d["Y_offset"]), float(d["Z_offset"])],
            [float(d["X_scale"]), float(d["Y_scale"]), float(d["Z_scale"])])


def load_rays(json_files, scene_loc=None, img_downscale=1.0, cache_dir=None, device="cuda", verbose=False):
    """The ray half of SatelliteDataset.load_data (datasets/satellite.py:406-481) with the RPC localisation on the GPU.
    json_files: metadata JSON paths (one per image, their order defines the image index); scene_loc: path of scene.loc_utm,
    an (offset, scale) pair, or None to derive it from these rays as init_scaling_params does.  Rays of an image found in
    cache_dir are read from there, others are generated and written there in the reference's format.
    Cache semantics follow :441-476: an 8-column cache holds UN-normalised rays (sun direction appended and normalised here);
    an 11-column cache holds FINAL rays (already normalised, sun direction included: recompute=False) and is passed through
    unchanged.  The reference takes its `recompute` flag from the last image only, so a mix of 8- and 11-column caches
    normalises either everything twice or nothing; such a mix is rejected here with a ValueError instead.
    Returns (all_rays fp32 [N,11] normalised, all_ids_img int64 [N], all_img_shapes [[h,w],...], scene_loc (offset, scale))."""
    import json
    import os
    import torch
    dev = torch.device(device)
    raws, ids, shapes, final = [], [], [], []
    for t, jp in enumerate(json_files):
        with open(jp) as f:
            d = json.load(f)
        img_id = os.path.splitext(os.path.basename(d["img"]))[0]                     # sat_utils.get_file_id
        h, w = int(d["height"] // img_downscale), int(d["width"] // img_downscale)
        cache_path = None if cache_dir is None else os.path.join(cache_dir, img_id + ".data")
        if cache_path is not None and os.path.exists(cache_path):
            raw = torch.load(cache_path, map_location="cpu")
            raw = torch.as_tensor(raw).to(dev)
            if raw.shape[0] != h * w or raw.shape[1] not in (8, 11):
                raise ValueError(f"{cache_path}: expected [{h * w}, 8|11] rays, found {tuple(raw.shape)}")
        else:
            raw = generate_rays(d["rpc"], float(d["min_alt"]), float(d["max_alt"]), h=h, w=w, img_downscale=img_downscale, device=dev)
            if cache_path is not None:
                os.makedirs(os.path.dirname(cache_path) or ".", exist_ok=True)
                torch.save(raw.cpu(), cache_path)
        final.append(raw.shape[1] == 11)      # recompute = False (:443-444): final normalised rays, taken as they are
        if raw.shape[1] == 8:      # sun directions are appended to freshly generated / 8-column cached rays (:455-458)
            sun = torch.tensor(sun_direction(d["sun_elevation"], d["sun_azimuth"]), dtype=torch.float64, device=dev)
            raw = torch.cat([raw.to(torch.float64), sun.expand(raw.shape[0], 3)], dim=1)
        raws.append(raw.to(torch.float64))
        ids.append(torch.full((raw.shape[0],), t, dtype=torch.int64, device=dev))
        shapes.append([h, w])
        if verbose:
            print(f"Image {img_id} loaded ( {t + 1} / {len(json_files)} )")
    all_raw = torch.cat(raws, 0)
    if any(final):
        if not all(final):
            raise ValueError("ray caches mix 8-column (un-normalised) and 11-column (final) files; regenerate one kind")
        if isinstance(scene_loc, (str, os.PathLike)):
            scene_loc = read_scene_loc(scene_loc)
        return all_raw.to(torch.float32), torch.cat(ids, 0), shapes, scene_loc
    if scene_loc is None:
        dloc = scene_loc_from_rays(all_raw)
        scene_loc = ([dloc["X_offset"], dloc["Y_offset"], dloc["Z_offset"]], [dloc["X_scale"], dloc["Y_scale"], dloc["Z_scale"]])
    elif isinstance(scene_loc, (str, os.PathLike)):
        scene_loc = read_scene_loc(scene_loc)
    all_rays = normalize_rays(all_raw, scene_loc[0], scene_loc[1]).to(torch.float32)
    return all_rays, torch.cat(ids, 0), shapes, scene_loc


def write_scene_loc(path, scene_loc_dict):
    """sat_utils.write_dict_to_json of init_scaling_params (:403)."""
    import json
    with open(path, "w") as f:
        json.dump(scene_loc_dict, f, indent=2)


def get_utmalt_from_nerf_prediction(rays, depth, scene_offset, scene_scale, double=True):
    """SatelliteDataset.get_utmalt_from_nerf_prediction (datasets/satellite.py:502-533, UTM branch) on the tensors' device:
    rays [N,>=6] normalised, depth [N,1] rendered depth -> (easts, norths, alts); fp64 by default, as the reference (the
    altitude of the DSM criterion: within 1 cm of the reference path)."""
    import torch
    if double:
        rays, depth = rays.double(), depth.double()
    off = torch.as_tensor(scene_offset, dtype=torch.float32).to(rays.device, rays.dtype)      # fp32 tensors in the dataset (:303-307)
    sc = torch.as_tensor(scene_scale, dtype=torch.float32).to(rays.device, rays.dtype)
    xyz = (rays[:, 0:3] + rays[:, 3:6] * depth.view(-1, 1)) * sc + off
    return xyz[:, 0], xyz[:, 1], xyz[:, 2]
