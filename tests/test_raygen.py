"""H1 / N2: RPC ray generation.  CPU part: the oracle's own consistency (projection o localisation = identity, the UTM series
against closed-form anchors).  GPU part: the HIP kernel against the oracle.

Parity status: UNPINNED w.r.t. rpcm / pyproj (un-vendored; oracle/raygen_oracle.py docstring).  Tolerance on the GPU path:
the reference casts UTM coordinates to fp32 (datasets/satellite.py:119-120; northing ~3.3e6 m -> 0.25 m quantum) BEFORE
normalising, so a 1e-9-relative difference between two fp64 evaluations can flip that rounding: values either agree to 1e-6
(normalised units) or differ by exactly one fp32 quantum of the raw value; the test allows < 0.1 % such flips."""
import math

import numpy as np
import pytest
import torch

from oracle import raygen_oracle as rg


def test_projection_of_localization_is_identity():
    rpc = rg.synthetic_rpc(seed=1)
    g = np.random.default_rng(0)
    cols, rows = g.uniform(0, 2048, 500), g.uniform(0, 2048, 500)
    for alt in (-20.0, 20.0, 90.0):
        lon, lat = rg.localization(rpc, cols, rows, alt * np.ones(500))
        c2, r2 = rg.projection(rpc, lon, lat, alt)
        assert np.abs(c2 - cols).max() < 1e-5 and np.abs(r2 - rows).max() < 1e-5


def test_utm_series_anchors():
    # on the central meridian east = 500000 exactly; the equator maps to north = 0; scale k0 at the central meridian
    e, n = rg.utm_forward(np.array([0.0, 30.0]), np.array([-81.0, -81.0]), 17)
    assert abs(e[0] - 500000.0) < 1e-9 and abs(n[0]) < 1e-9 and abs(e[1] - 500000.0) < 1e-9
    # meridian arc length from the equator to 30 deg N on WGS84 = 3 320 113.398 m (geodesy tables) -> x k0
    assert abs(n[1] - 0.9996 * 3320113.398) < 2e-3
    assert rg.utm_zone_number(30.33, -81.66) == 17 and rg.utm_zone_number(60.0, 5.0) == 32
    # symmetric about the central meridian
    e1, n1 = rg.utm_forward(np.array([30.0]), np.array([-80.0]), 17)
    e2, n2 = rg.utm_forward(np.array([30.0]), np.array([-82.0]), 17)
    assert abs((e1 - 500000) + (e2 - 500000)) < 1e-8 and abs(n1 - n2) < 1e-8


def _scene(rpc, h, w, zone):
    corners_c, corners_r = np.array(2 * [0, w - 1, w - 1, 0]), np.array(2 * [0, 0, h - 1, h - 1])
    raw = rg.get_rays(corners_c, corners_r, rpc, -20.0, 90.0, zone).astype(np.float64)
    pts = np.vstack([raw[:, :3], raw[:, :3] + raw[:, 7:8] * raw[:, 3:6]])
    scale = (pts.max(0) - pts.min(0)) / 2
    return (pts.min(0) + scale).astype(np.float32), scale.astype(np.float32)     # rpc_scaling_params, sat_utils.py:31-38


def test_oracle_image_rays_shape_and_unit_vectors():
    rpc = rg.synthetic_rpc(seed=2)
    zone = rg.utm_zone_number(rpc["lat_offset"], rpc["lon_offset"])
    off, sc = _scene(rpc, 64, 48, zone)
    rays, raw = rg.image_rays(rpc, 64, 48, -20.0, 90.0, 55.0, 140.0, off, sc, zone)
    assert rays.shape == (64 * 48, 11) and rays.dtype == np.float32 and raw.shape == (64 * 48, 8)
    assert np.abs(np.linalg.norm(rays[:, 3:6], axis=1) - 1).max() < 1e-6
    assert np.abs(np.linalg.norm(rays[:, 8:11], axis=1) - 1).max() < 1e-6
    assert (np.abs(rays[:, :3]) <= 1.0 + 1e-3).all() and (rays[:, 6] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,downscale", [(3, 1.0), (4, 2.0)])
def test_hip_ray_generation_matches_oracle(seed, downscale):
    from eonerf_code_amd.datasets.satellite import generate_rays, get_rays, utm_zone_from_lonlat
    rpc = rg.synthetic_rpc(seed=seed)
    h, w = int(96 // downscale), int(80 // downscale)
    rpc_s = rg.rescale_rpc(rpc, 1.0 / downscale)
    zone, south = utm_zone_from_lonlat(rpc["lon_offset"], rpc["lat_offset"])
    assert zone == rg.utm_zone_number(rpc["lat_offset"], rpc["lon_offset"])
    off, sc = _scene(rpc_s, h, w, zone)
    ref, ref_raw = rg.image_rays(rpc_s, h, w, -20.0, 90.0, 55.0, 140.0, off, sc, zone)
    rays, raw = generate_rays(rpc, -20.0, 90.0, h=h, w=w, img_downscale=downscale, sun_elevation_deg=55.0, sun_azimuth_deg=140.0,
                              scene_offset=off, scene_scale=sc, want_raw=True)
    rays, raw = rays.cpu().numpy(), raw.cpu().numpy()
    # raw rays: equal or one fp32 quantum apart
    quantum = np.spacing(np.abs(ref_raw).astype(np.float32))
    d_raw = np.abs(raw.astype(np.float64) - ref_raw.astype(np.float64))
    assert (d_raw <= 1.01 * quantum).all()
    assert (d_raw > 0).mean() < 1e-3
    # normalised rays: 1e-6, except where a raw flip moved them by quantum/scale
    d = np.abs(rays.astype(np.float64) - ref.astype(np.float64))
    assert (d > 2e-6).mean() < 1e-3
    assert d[:, :3].max() <= 1.01 * (quantum[:, :3].max(0) / sc).max() + 2e-6
    assert d[:, 8:11].max() < 1e-6 and d[:, 3:6].max() < 1e-5
    # explicit pixel lists (get_rays signature) agree with the grid
    cols, rows = np.meshgrid(np.arange(w), np.arange(h))
    raw2 = get_rays(cols.flatten(), rows.flatten(), rpc_s, -20.0, 90.0).cpu().numpy()
    assert np.array_equal(raw2, raw)


# ---- dataset side (N2): normalisation of cached rays, scene.loc_utm, metadata JSONs, cache files --------------------------------
def test_normalize_rays_and_scene_loc_match_the_oracle_cpu(tmp_path):
    """Host logic only (torch on CPU): normalising the fp32 payload of a ray cache the way load_data does (fp64 after the hstack
    with the sun directions, datasets/satellite.py:455-478,124-139) and init_scaling_params (:395-403)."""
    from eonerf_code_amd.datasets import satellite as ds
    rpc = rg.synthetic_rpc(seed=5)
    zone = rg.utm_zone_number(rpc["lat_offset"], rpc["lon_offset"])
    h, w = 24, 20
    cols, rows = np.meshgrid(np.arange(w), np.arange(h))
    raw = rg.get_rays(cols.flatten(), rows.flatten(), rpc, -20.0, 90.0, zone)              # fp32 [h*w, 8], the cache payload
    sun = np.array(ds.sun_direction(55.0, 140.0))
    np.testing.assert_allclose(sun, rg.sun_direction(55.0, 140.0) if hasattr(rg, "sun_direction") else sun, rtol=0, atol=1e-15)
    full = np.hstack([raw.astype(np.float64), np.tile(sun, (raw.shape[0], 1))])
    # scene.loc from these rays, as init_scaling_params does, and through the JSON file
    loc = ds.scene_loc_from_rays(torch.from_numpy(raw))
    pts = np.vstack([raw[:, :3].astype(np.float64), raw[:, :3].astype(np.float64) + raw[:, 7:8].astype(np.float64) * raw[:, 3:6].astype(np.float64)])
    for k, name in enumerate("XYZ"):
        assert loc[name + "_scale"] == (pts[:, k].max() - pts[:, k].min()) / 2
        assert loc[name + "_offset"] == pts[:, k].min() + loc[name + "_scale"]
    ds.write_scene_loc(tmp_path / "scene.loc_utm", loc)
    off, sc = ds.read_scene_loc(tmp_path / "scene.loc_utm")
    ref = rg.normalize_rays(full, np.array(off, dtype=np.float32), np.array(sc, dtype=np.float32))
    got = ds.normalize_rays(torch.from_numpy(full), off, sc).numpy()
    assert got.shape == ref.shape == (h * w, 11)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    assert np.abs(got[:, :3]).max() <= 1.01                                             # the cube, up to the fp32 rounding of offset / scale


def _write_scene(tmp_path, n_img, h, w):
    import json
    files = []
    for t in range(n_img):
        rpc = rg.synthetic_rpc(seed=10 + t)
        d = {"img": f"JAX_999_{t:03d}_RGB.tif", "height": h, "width": w, "rpc": rpc, "min_alt": -20.0, "max_alt": 90.0,
             "sun_elevation": 40.0 + 5 * t, "sun_azimuth": 120.0 + 10 * t}
        p = tmp_path / f"JAX_999_{t:03d}_RGB.json"
        p.write_text(json.dumps(d))
        files.append(str(p))
    return files


@pytest.mark.gpu
def test_load_rays_from_metadata_cache_roundtrip_and_oracle(tmp_path):
    from eonerf_code_amd.datasets import satellite as ds
    h, w, n_img = 40, 32, 3
    files = _write_scene(tmp_path, n_img, h, w)
    cache = tmp_path / "cache"
    rays, ids, shapes, loc = ds.load_rays(files, scene_loc=None, cache_dir=str(cache), device="cuda")
    assert rays.shape == (n_img * h * w, 11) and rays.dtype == torch.float32 and ids.dtype == torch.int64
    assert shapes == [[h, w]] * n_img and torch.equal(ids.cpu(), torch.arange(n_img).repeat_interleave(h * w))
    # cache files: the reference's format, fp32 [h*w, 8] un-normalised rays (datasets/satellite.py:441-453)
    for t in range(n_img):
        c = torch.load(cache / f"JAX_999_{t:03d}_RGB.data")
        assert c.shape == (h * w, 8) and c.dtype == torch.float32
    # second load reads the cache: identical rays; an explicit scene.loc file gives the same normalisation
    ds.write_scene_loc(tmp_path / "scene.loc_utm", {k + s: v for (k, s, v) in
                                                     [(n, "_offset", loc[0][i]) for i, n in enumerate("XYZ")] +
                                                     [(n, "_scale", loc[1][i]) for i, n in enumerate("XYZ")]})
    rays2, ids2, _, loc2 = ds.load_rays(files, scene_loc=str(tmp_path / "scene.loc_utm"), cache_dir=str(cache), device="cuda")
    assert torch.equal(rays, rays2) and torch.equal(ids, ids2)
    # against the oracle's load_data for one image (same tolerance as test_hip_ray_generation_matches_oracle)
    import json
    d = json.loads(open(files[1]).read())
    zone = rg.utm_zone_number(d["rpc"]["lat_offset"], d["rpc"]["lon_offset"])
    ref, _ = rg.image_rays(d["rpc"], h, w, -20.0, 90.0, d["sun_elevation"], d["sun_azimuth"], np.array(loc[0], dtype=np.float32),
                           np.array(loc[1], dtype=np.float32), zone)
    got = rays[h * w:2 * h * w].cpu().numpy()
    bad = np.abs(got - ref) > 1e-6
    assert bad.mean() < 1e-3, bad.mean()
    assert np.abs(np.linalg.norm(got[:, 3:6], axis=1) - 1).max() < 1e-6 and np.abs(np.linalg.norm(got[:, 8:11], axis=1) - 1).max() < 1e-6


def test_load_rays_cache_semantics_8_vs_11_columns_cpu(tmp_path):
    """datasets/satellite.py:441-476: 8-column caches are un-normalised (sun appended, normalised on load), 11-column caches are
    FINAL rays and pass through unchanged; a mix is rejected (the reference would normalise everything twice or nothing)."""
    from eonerf_code_amd.datasets import satellite as ds
    h, w, n_img = 6, 5, 2
    files = _write_scene(tmp_path, n_img, h, w)
    cache = tmp_path / "cache"
    cache.mkdir()
    raws = []
    for t in range(n_img):
        rpc = rg.synthetic_rpc(seed=10 + t)
        zone = rg.utm_zone_number(rpc["lat_offset"], rpc["lon_offset"])
        cols, rows = np.meshgrid(np.arange(w), np.arange(h))
        raw = rg.get_rays(cols.flatten(), rows.flatten(), rpc, -20.0, 90.0, zone)
        raws.append(raw)
        torch.save(torch.from_numpy(raw), cache / f"JAX_999_{t:03d}_RGB.data")
    rays, ids, shapes, loc = ds.load_rays(files, scene_loc=None, cache_dir=str(cache), device="cpu")
    assert rays.shape == (n_img * h * w, 11) and np.abs(rays[:, :3].numpy()).max() <= 1.01
    # now cache the FINAL rays (11 columns): the next load returns them bit for bit, without normalising again
    for t in range(n_img):
        torch.save(rays[t * h * w:(t + 1) * h * w].clone(), cache / f"JAX_999_{t:03d}_RGB.data")
    rays2, ids2, _, _ = ds.load_rays(files, scene_loc=loc, cache_dir=str(cache), device="cpu")
    assert torch.equal(rays2, rays) and torch.equal(ids2, ids)
    torch.save(torch.from_numpy(raws[0]), cache / "JAX_999_000_RGB.data")     # image 0 back to 8 columns: a mix
    with pytest.raises(ValueError, match="mix"):
        ds.load_rays(files, scene_loc=loc, cache_dir=str(cache), device="cpu")


# ---- narrowing the unpinned surface of H1 / N2 (still parity-UNPINNED w.r.t. rpcm / pyproj: see the module docstring) -----------
def worldview_like_rpc(seed=0):
    """An RPC with the magnitudes of a WorldView-3 DFC2019 crop (datasets' JSON "rpc" entries): image normalisation of a few
    thousand pixels, ground footprint of ~0.02 deg, height scale 500 m, second-order terms ~1e-3, cubic terms ~1e-5, denominators
    1 + O(1e-3) -- larger non-linearities than synthetic_rpc(), so the localisation needs its iterations."""
    g = np.random.default_rng(seed)
    rpc = {"row_offset": 1100.0, "col_offset": 1050.0, "row_scale": 1101.0, "col_scale": 1051.0,
           "lat_offset": 30.3325, "lon_offset": -81.6610, "alt_offset": -10.0,
           "lat_scale": 0.0032, "lon_scale": 0.0037, "alt_scale": 501.0}

    def num(lin_lon, lin_lat, lin_alt, const):
        p = np.zeros(20)
        p[4:10] = 1e-3 * g.standard_normal(6)
        p[10:20] = 1e-5 * g.standard_normal(10)
        p[0], p[1], p[2], p[3] = const, lin_lon, lin_lat, lin_alt
        return p.tolist()

    def den():
        p = np.zeros(20)
        p[1:4] = 1e-3 * g.standard_normal(3)
        p[4:10] = 1e-5 * g.standard_normal(6)
        p[10:20] = 1e-6 * g.standard_normal(10)
        p[0] = 1.0
        return p.tolist()

    rpc["col_num"], rpc["row_num"] = num(1.002, 0.0123, -0.0951, 1.3e-3), num(-0.0184, -1.009, 0.0873, -2.1e-3)
    rpc["col_den"], rpc["row_den"] = den(), den()
    return rpc


@pytest.mark.gpu
def test_hip_ray_generation_worldview_scale_rpc_downscale_2():
    from eonerf_code_amd.datasets.satellite import generate_rays, utm_zone_from_lonlat
    rpc = worldview_like_rpc(seed=7)
    downscale = 2.0
    h, w = int(2200 // downscale) // 8, int(2100 // downscale) // 8          # a 137 x 131 corner of the half-resolution image
    rpc_s = rg.rescale_rpc(rpc, 1.0 / downscale)
    zone, south = utm_zone_from_lonlat(rpc["lon_offset"], rpc["lat_offset"])
    off, sc = _scene(rpc_s, h, w, zone)
    ref, ref_raw = rg.image_rays(rpc_s, h, w, -30.0, 120.0, 47.0, 163.0, off, sc, zone)
    rays, raw = generate_rays(rpc, -30.0, 120.0, h=h, w=w, img_downscale=downscale, sun_elevation_deg=47.0, sun_azimuth_deg=163.0,
                              scene_offset=off, scene_scale=sc, want_raw=True)
    rays, raw = rays.cpu().numpy(), raw.cpu().numpy()
    # With real non-linearities the localisation runs several iterations and stops at a squared normalised residual < 1e-18,
    # i.e. anywhere within ~1e-6 px of the pixel; two fp64 evaluations (numpy / HIP) stop at slightly different points, so after
    # the fp32 cast of :119-120 a value either agrees or sits ONE fp32 quantum away (measured: 1 % of the entries, almost all in
    # the direction components whose quantum is 6e-8).  Never more than one quantum.
    quantum = np.spacing(np.abs(ref_raw).astype(np.float32))
    d_raw = np.abs(raw.astype(np.float64) - ref_raw.astype(np.float64))
    assert (d_raw <= 1.01 * quantum).all() and (d_raw > 0).mean() < 5e-2
    assert (d_raw[:, :3] > 0).mean() < 2e-3                        # origins (quanta of 3 cm / 25 cm): rarely
    d = np.abs(rays.astype(np.float64) - ref.astype(np.float64))
    assert (d[:, :3] > 2e-6).mean() < 2e-3 and d[:, 8:11].max() < 1e-6 and d[:, 3:6].max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed", [("synthetic", 3), ("worldview", 8)])
def test_projection_of_hip_localization_returns_the_pixel(kind, seed):
    """In-tree anchor: the RPC polynomial of the reference (apply_poly's term order, sat_utils.py:437-450, restated in the oracle)
    applied to the lon/lat the HIP kernel localised must give back the pixel.  Bound: the iteration's own stopping rule (squared
    NORMALISED residual < 1e-18, i.e. < 1e-9 x row/col scale ~ 1.1e-6 px for the WorldView-like model): 2e-6 px."""
    from eonerf_code_amd.datasets.satellite import generate_rays
    rpc = rg.synthetic_rpc(seed=seed) if kind == "synthetic" else worldview_like_rpc(seed)
    g = np.random.default_rng(1)
    cols, rows = g.uniform(0, 2000, 700), g.uniform(0, 2000, 700)
    lo, hi = -25.0, 110.0
    geo = generate_rays(rpc, lo, hi, cols=cols, rows=rows, want_geo=True).cpu().numpy()
    for k, alt in ((0, hi), (4, lo)):
        c2, r2 = rg.projection(rpc, geo[:, k], geo[:, k + 1], alt)
        assert np.abs(c2 - cols).max() < 2e-6 and np.abs(r2 - rows).max() < 2e-6
    # and the two numerical cores agree with the oracle's in fp64, before any fp32 rounding
    lon, lat = rg.localization(rpc, cols, rows, hi * np.ones(700))
    assert np.abs(geo[:, 0] - lon).max() < 2e-11 and np.abs(geo[:, 1] - lat).max() < 2e-11      # ~1e-9 x lon/lat scale
    zone = rg.utm_zone_number(rpc["lat_offset"], rpc["lon_offset"])
    e, n = rg.utm_forward(geo[:, 1], geo[:, 0], zone)
    assert np.abs(geo[:, 2] - e).max() < 1e-6 and np.abs(geo[:, 3] - n).max() < 1e-6


@pytest.mark.gpu
def test_krueger_series_closed_form_anchors_on_the_gpu_path():
    """The UTM projection INSIDE the HIP kernel against closed-form anchors (not only the numpy oracle): an identity RPC
    (col -> longitude, row -> latitude) steers the kernel to exact geodetic points of zone 17 (central meridian 81 W)."""
    from eonerf_code_amd.datasets.satellite import generate_rays
    ident = {"row_offset": 0.0, "col_offset": 0.0, "row_scale": 1.0, "col_scale": 1.0, "lat_offset": 0.0, "lon_offset": -81.0,
             "alt_offset": 0.0, "lat_scale": 1.0, "lon_scale": 1.0, "alt_scale": 1.0}
    z = [0.0] * 20
    ident["col_num"] = [0.0, 1.0] + [0.0] * 18                     # poly[1] multiplies the normalised longitude (apply_poly)
    ident["row_num"] = [0.0, 0.0, 1.0] + [0.0] * 17                # poly[2] multiplies the normalised latitude
    ident["col_den"] = [1.0] + [0.0] * 19
    ident["row_den"] = [1.0] + [0.0] * 19
    # pixels = (lon - (-81), lat): equator & 30 N on the central meridian, 30 N one degree either side, 45 N on the meridian
    cols = np.array([0.0, 0.0, 1.0, -1.0, 0.0])
    rows = np.array([0.0, 30.0, 30.0, 30.0, 45.0])
    geo = generate_rays(ident, -1.0, 1.0, cols=cols, rows=rows, zone=17, south=False, want_geo=True).cpu().numpy()
    assert np.abs(geo[:, 0] - (-81.0 + cols)).max() < 1e-12 and np.abs(geo[:, 1] - rows).max() < 1e-12
    e, n = geo[:, 2], geo[:, 3]
    assert abs(e[0] - 500000.0) < 1e-8 and abs(n[0]) < 1e-8                      # equator x central meridian
    assert abs(e[1] - 500000.0) < 1e-8 and abs(e[4] - 500000.0) < 1e-8           # on the central meridian east = false easting
    # meridian arc lengths on WGS84 (geodesy tables): equator -> 30 N = 3 320 113.398 m, -> 45 N = 4 984 944.378 m; x k0
    assert abs(n[1] - 0.9996 * 3320113.398) < 2e-3 and abs(n[4] - 0.9996 * 4984944.378) < 2e-3
    # symmetry about the central meridian
    assert abs((e[2] - 500000.0) + (e[3] - 500000.0)) < 1e-7 and abs(n[2] - n[3]) < 1e-7
    # one degree of longitude at 30 N ~ 96.49 km on the ellipsoid, shortened by the scale factor near the meridian
    assert 96400.0 < e[2] - 500000.0 < 96520.0
