"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy fp64) of RPC ray generation.

Follows datasets/satellite.py:57-139,456-458,486-500 (get_rays, normalize_rays, get_dir_vec_from_el_az,
get_sun_dirs) and sat_utils.py:41-59,99-116,437-450 (rescale_rpc, utm_from_latlon, apply_poly) of the reference.

PARITY STATUS: **unpinned**.  The two numerical cores live in un-vendored third-party packages that are absent from
/root/reference and from this image, and the reference ships no fixture for them:
  * rpcm (requirements.txt:8, unpinned) RPCModel.localization -> restated below from its published iterative
    algorithm (localization_iterative: a base of two EPS steps in normalised lon/lat is projected into the image and the
    residual is decomposed on it; EPS = 2 for the first iteration, 0.1 afterwards; stop when the squared normalised
    image residual is < 1e-18).  The polynomial term order is pinned by the in-tree copy sat_utils.py:437-450.
  * pyproj 3.0.1 / PROJ (setup_env.sh:11) "+proj=utm": restated as the 6th-order Krueger series (Karney 2011), the
    algorithm PROJ's etmerc implements; agreement with PROJ is at the nanometre level by construction of the series.
In-tree anchors used by tests/test_raygen.py: apply_poly term order, the normalisation / sun-direction formulas, the
fp32 round trip at datasets/satellite.py:119-120, and the property projection(localization(x)) == x.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------- RPC (rpcm dict format)
def apply_poly(poly, x, y, z):
    """sat_utils.py:437-450 (x = normalised lat, y = normalised lon, z = normalised alt)."""
    out = 0
    out += poly[0]
    out += poly[1] * y + poly[2] * x + poly[3] * z
    out += poly[4] * y * x + poly[5] * y * z + poly[6] * x * z
    out += poly[7] * y * y + poly[8] * x * x + poly[9] * z * z
    out += poly[10] * x * y * z
    out += poly[11] * y * y * y
    out += poly[12] * y * x * x + poly[13] * y * z * z + poly[14] * y * y * x
    out += poly[15] * x * x * x
    out += poly[16] * x * z * z + poly[17] * y * y * z + poly[18] * x * x * z
    out += poly[19] * z * z * z
    return out


def apply_rfm(num, den, x, y, z):
    return apply_poly(num, x, y, z) / apply_poly(den, x, y, z)


def rescale_rpc(rpc, alpha):
    """sat_utils.py:41-59."""
    r = dict(rpc)
    for k in ("row_scale", "col_scale", "row_offset", "col_offset"):
        r[k] = rpc[k] * float(alpha)
    return r


def projection(rpc, lon, lat, alt):
    nlon = (lon - rpc["lon_offset"]) / rpc["lon_scale"]
    nlat = (lat - rpc["lat_offset"]) / rpc["lat_scale"]
    nalt = (alt - rpc["alt_offset"]) / rpc["alt_scale"]
    col = apply_rfm(rpc["col_num"], rpc["col_den"], nlat, nlon, nalt) * rpc["col_scale"] + rpc["col_offset"]
    row = apply_rfm(rpc["row_num"], rpc["row_den"], nlat, nlon, nalt) * rpc["row_scale"] + rpc["row_offset"]
    return col, row


def localization(rpc, col, row, alt, max_iter=100):
    """rpcm RPCModel.localization_iterative restated (see module docstring)."""
    col, row, alt = (np.asarray(v, dtype=np.float64) for v in (col, row, alt))
    ncol = (col - rpc["col_offset"]) / rpc["col_scale"]
    nrow = (row - rpc["row_offset"]) / rpc["row_scale"]
    nalt = (alt - rpc["alt_offset"]) / rpc["alt_scale"]
    lon = -np.ones_like(ncol)
    lat = -np.ones_like(ncol)
    eps = 2.0
    cn, cd, rn, rd = rpc["col_num"], rpc["col_den"], rpc["row_num"], rpc["row_den"]
    x0, y0 = apply_rfm(cn, cd, lat, lon, nalt), apply_rfm(rn, rd, lat, lon, nalt)
    x1, y1 = apply_rfm(cn, cd, lat, lon + eps, nalt), apply_rfm(rn, rd, lat, lon + eps, nalt)
    x2, y2 = apply_rfm(cn, cd, lat + eps, lon, nalt), apply_rfm(rn, rd, lat + eps, lon, nalt)
    n = 0
    while not np.all((x0 - ncol) ** 2 + (y0 - nrow) ** 2 < 1e-18):
        if n > max_iter:
            raise RuntimeError("max localization iterations exceeded")
        e1x, e1y, e2x, e2y = x1 - x0, y1 - y0, x2 - x0, y2 - y0
        ux, uy = ncol - x0, nrow - y0
        a1 = (ux * e1x + uy * e1y) / (e1x * e1x + e1y * e1y)
        a2 = (ux * e2x + uy * e2y) / (e2x * e2x + e2y * e2y)
        lon = lon + a1 * eps
        lat = lat + a2 * eps
        eps = 0.1
        x0, y0 = apply_rfm(cn, cd, lat, lon, nalt), apply_rfm(rn, rd, lat, lon, nalt)
        x1, y1 = apply_rfm(cn, cd, lat, lon + eps, nalt), apply_rfm(rn, rd, lat, lon + eps, nalt)
        x2, y2 = apply_rfm(cn, cd, lat + eps, lon, nalt), apply_rfm(rn, rd, lat + eps, lon, nalt)
        n += 1
    return lon * rpc["lon_scale"] + rpc["lon_offset"], lat * rpc["lat_scale"] + rpc["lat_offset"]


# ----------------------------------------------------------------------------- UTM (PROJ +proj=utm, WGS84)
WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
UTM_K0 = 0.9996


def utm_zone_number(lat, lon):
    """utm.latlon_to_zone_number (incl. the Norway / Svalbard exceptions)."""
    if 56 <= lat < 64 and 3 <= lon < 12:
        return 32
    if 72 <= lat <= 84 and lon >= 0:
        if lon < 9:
            return 31
        if lon < 21:
            return 33
        if lon < 33:
            return 35
        if lon < 42:
            return 37
    return int((lon + 180) / 6) % 60 + 1


def krueger_alpha():
    n = WGS84_F / (2.0 - WGS84_F)
    n2, n3, n4, n5, n6 = n ** 2, n ** 3, n ** 4, n ** 5, n ** 6
    A = WGS84_A / (1 + n) * (1 + n2 / 4 + n4 / 64 + n6 / 256)
    alpha = [n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800,
             13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360,
             61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440,
             49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600,
             34729 * n5 / 80640 - 3418889 * n6 / 1995840,
             212378941 * n6 / 319334400]
    e = math.sqrt(WGS84_F * (2 - WGS84_F))
    return A, alpha, e


def utm_forward(lats, lons, zone, south=False):
    """(east, north) of "+proj=utm +zone=<zone> [+south]" (sat_utils.py:99-116 builds exactly this projection)."""
    A, alpha, e = krueger_alpha()
    phi = np.radians(np.asarray(lats, dtype=np.float64))
    lam = np.radians(np.asarray(lons, dtype=np.float64) - (zone * 6.0 - 183.0))
    s = np.sin(phi)
    t = np.sinh(np.arctanh(s) - e * np.arctanh(e * s))
    xi_p = np.arctan2(t, np.cos(lam))
    eta_p = np.arctanh(np.sin(lam) / np.sqrt(1 + t * t))
    xi, eta = xi_p.copy(), eta_p.copy()
    for j, aj in enumerate(alpha, start=1):
        xi = xi + aj * np.sin(2 * j * xi_p) * np.cosh(2 * j * eta_p)
        eta = eta + aj * np.cos(2 * j * xi_p) * np.sinh(2 * j * eta_p)
    east = 500000.0 + UTM_K0 * A * eta
    north = UTM_K0 * A * xi + (10000000.0 if south else 0.0)
    return east, north


# ----------------------------------------------------------------------------- rays
def get_dir_vec_from_el_az(elevation_deg, azimuth_deg):
    """datasets/satellite.py:57-63."""
    el, az = np.radians(90 - elevation_deg), np.radians(azimuth_deg)
    return -1.0 * np.array([np.sin(az) * np.cos(el), np.cos(az) * np.cos(el), np.sin(el)])


def get_rays(cols, rows, rpc, min_alt, max_alt, zone, south=False):
    """datasets/satellite.py:65-121 (utm branch): -> float32 [N,8] (o3, d3, near, far) incl. the fp32 cast at :119-120."""
    cols, rows = np.asarray(cols, dtype=np.float64), np.asarray(rows, dtype=np.float64)
    max_alts, min_alts = float(max_alt) * np.ones(cols.shape), float(min_alt) * np.ones(cols.shape)
    lons, lats = localization(rpc, cols, rows, max_alts)
    e, n = utm_forward(lats, lons, zone, south)
    near = np.vstack([e, n, max_alts]).T
    lons, lats = localization(rpc, cols, rows, min_alts)
    e, n = utm_forward(lats, lons, zone, south)
    far = np.vstack([e, n, min_alts]).T
    d = far - near
    fars = np.linalg.norm(d, axis=1)
    rays = np.hstack([near, d / fars[:, None], np.zeros((len(cols), 1)), fars[:, None]])
    return rays.astype(np.float32)


def normalize_rays(rays, scene_offset, scene_scale):
    """datasets/satellite.py:124-139; rays float64 [N,11] holding the fp32 values of get_rays + fp64 sun dirs,
    scene_offset / scene_scale float32 arrays (the dataset keeps them as fp32 torch tensors, :303-307)."""
    rays = np.array(rays, dtype=np.float64)
    off, sc = np.asarray(scene_offset, dtype=np.float32), np.asarray(scene_scale, dtype=np.float32)
    o = rays[:, :3]
    e = rays[:, :3] + rays[:, 3:6] * rays[:, 7:8]
    o_n, e_n = (o - off) / sc, (e - off) / sc
    d = e_n - o_n
    fars = np.linalg.norm(d, axis=1)
    out = np.hstack([o_n, d / fars[:, None], np.zeros((len(o), 1)), fars[:, None]])
    sun = rays[:, 8:11] / sc
    sun = sun / np.linalg.norm(sun, axis=1)[:, None]
    return np.hstack([out, sun])


def image_rays(rpc, h, w, min_alt, max_alt, sun_elevation_deg, sun_azimuth_deg, scene_offset, scene_scale, zone, south=False,
               cols=None, rows=None):
    """load_data for one image (datasets/satellite.py:447-478): float32 [h*w, 11] normalised rays."""
    if cols is None:
        cols, rows = np.meshgrid(np.arange(w), np.arange(h))
        cols, rows = cols.flatten(), rows.flatten()
    raw = get_rays(cols, rows, rpc, min_alt, max_alt, zone, south)
    sun = np.tile(get_dir_vec_from_el_az(90 - float(sun_elevation_deg), float(sun_azimuth_deg)), (raw.shape[0], 1))   # :457
    rays = np.hstack([raw, sun])                       # float32 | float64 -> float64 (:458)
    return normalize_rays(rays, scene_offset, scene_scale).astype(np.float32), raw


def synthetic_rpc(seed=0, lat0=30.33, lon0=-81.66, alt0=20.0, size=2048):
    """A well-conditioned synthetic RPC around Jacksonville (JAX-like): near-affine ground->image map (0.3 m GSD, off-nadir
    tilt through the altitude term) plus small higher-order terms and a non-trivial denominator."""
    g = np.random.default_rng(seed)
    half = size / 2.0
    rpc = {"row_offset": half, "col_offset": half, "row_scale": half, "col_scale": half,
           "lat_offset": lat0, "lon_offset": lon0, "alt_offset": alt0,
           "lat_scale": 0.3 * half / 111000.0 * 1.1, "lon_scale": 0.3 * half / (111000.0 * math.cos(math.radians(lat0))) * 1.1,
           "alt_scale": 120.0}

    def poly(lin_lon, lin_lat, lin_alt, const, scale):
        p = scale * g.standard_normal(20)
        p[0], p[1], p[2], p[3] = const, lin_lon, lin_lat, lin_alt
        return p.tolist()

    def den():
        p = 1e-4 * g.standard_normal(20)
        p[0] = 1.0
        return p.tolist()

    rpc["col_num"] = poly(1.05, 0.03, 0.02, 0.002, 2e-4)
    rpc["row_num"] = poly(-0.04, -1.04, 0.035, -0.003, 2e-4)
    rpc["col_den"], rpc["row_den"] = den(), den()
    return rpc
