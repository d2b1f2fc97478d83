// RPC ray generation on the device (H1 of SURVEY.md 8a): pixel (col,row) -> RPC localisation at max/min altitude ->
// UTM -> origin / unit direction / far -> fp32 round trip -> scene normalisation -> [N,11] fp32 rays.
// Follows datasets/satellite.py:57-139,456-458 and sat_utils.py:99-116,437-450 of the reference.  The two numerical
// cores belong to un-vendored packages and are restated from their published algorithms:
//   rpcm RPCModel.localization_iterative  (EPS-base iteration on the normalised rational cubic, stop at 1e-18)
//   PROJ "+proj=utm" (etmerc)             (6th-order Krueger series on WGS84)
// Everything is fp64 (one thread per pixel) except the reference's own fp32 cast (datasets/satellite.py:119-120).
#include <hip/hip_runtime.h>
#include "eonerf_raygen.h"

namespace {

__device__ __forceinline__ double poly20(const double* p, double x, double y, double z) {   // sat_utils.py:437-450
    double out = 0;
    out += p[0];
    out += p[1] * y + p[2] * x + p[3] * z;
    out += p[4] * y * x + p[5] * y * z + p[6] * x * z;
    out += p[7] * y * y + p[8] * x * x + p[9] * z * z;
    out += p[10] * x * y * z;
    out += p[11] * y * y * y;
    out += p[12] * y * x * x + p[13] * y * z * z + p[14] * y * y * x;
    out += p[15] * x * x * x;
    out += p[16] * x * z * z + p[17] * y * y * z + p[18] * x * x * z;
    out += p[19] * z * z * z;
    return out;
}
__device__ __forceinline__ void project_n(const RpcModel& r, double nlat, double nlon, double nalt, double& x, double& y) {
    x = poly20(r.col_num, nlat, nlon, nalt) / poly20(r.col_den, nlat, nlon, nalt);
    y = poly20(r.row_num, nlat, nlon, nalt) / poly20(r.row_den, nlat, nlon, nalt);
}

// rpcm localization_iterative for one point; returns lon/lat in degrees
__device__ void localize(const RpcModel& r, double col, double row, double alt, double& lon_deg, double& lat_deg) {
    const double ncol = (col - r.col_offset) / r.col_scale, nrow = (row - r.row_offset) / r.row_scale;
    const double nalt = (alt - r.alt_offset) / r.alt_scale;
    double lon = -1.0, lat = -1.0, eps = 2.0;
    double x0, y0, x1, y1, x2, y2;
    project_n(r, lat, lon, nalt, x0, y0);
    project_n(r, lat, lon + eps, nalt, x1, y1);
    project_n(r, lat + eps, lon, nalt, x2, y2);
    for (int n = 0; n <= 100; ++n) {
        const double ux = ncol - x0, uy = nrow - y0;
        if (ux * ux + uy * uy < 1e-18) break;
        const double e1x = x1 - x0, e1y = y1 - y0, e2x = x2 - x0, e2y = y2 - y0;
        const double a1 = (ux * e1x + uy * e1y) / (e1x * e1x + e1y * e1y);
        const double a2 = (ux * e2x + uy * e2y) / (e2x * e2x + e2y * e2y);
        lon += a1 * eps;
        lat += a2 * eps;
        eps = 0.1;
        project_n(r, lat, lon, nalt, x0, y0);
        project_n(r, lat, lon + eps, nalt, x1, y1);
        project_n(r, lat + eps, lon, nalt, x2, y2);
    }
    lon_deg = lon * r.lon_scale + r.lon_offset;
    lat_deg = lat * r.lat_scale + r.lat_offset;
}

// WGS84 transverse Mercator, 6th-order Krueger series (PROJ etmerc)
__device__ void utm_forward(const UtmParams& u, double lat_deg, double lon_deg, double& east, double& north) {
    const double d2r = 0.017453292519943295;
    const double phi = lat_deg * d2r, lam = (lon_deg - u.lon0_deg) * d2r;
    const double s = sin(phi);
    const double t = sinh(atanh(s) - u.e * atanh(u.e * s));
    const double xi_p = atan2(t, cos(lam));
    const double eta_p = atanh(sin(lam) / sqrt(1.0 + t * t));
    double xi = xi_p, eta = eta_p;
#pragma unroll
    for (int j = 1; j <= 6; ++j) {
        xi += u.alpha[j - 1] * sin(2 * j * xi_p) * cosh(2 * j * eta_p);
        eta += u.alpha[j - 1] * cos(2 * j * xi_p) * sinh(2 * j * eta_p);
    }
    east = 500000.0 + u.k0A * eta;
    north = u.k0A * xi + u.false_north;
}

__global__ __launch_bounds__(128) void k_raygen(RayGenArgs a) {
    const long i = (long)blockIdx.x * 128 + threadIdx.x;
    if (i >= a.n) return;
    double col, row;
    if (a.cols) { col = a.cols[i]; row = a.rows[i]; }
    else { col = (double)(i % a.width); row = (double)(i / a.width); }       // np.meshgrid(arange(w), arange(h)).flatten()
    double lon, lat, e0, n0, e1, n1;
    localize(a.rpc, col, row, a.max_alt, lon, lat);          // highest points are the closest to the camera (:87-91)
    utm_forward(a.utm, lat, lon, e0, n0);
    if (a.geo) { a.geo[i * 8] = lon; a.geo[i * 8 + 1] = lat; a.geo[i * 8 + 2] = e0; a.geo[i * 8 + 3] = n0; }
    localize(a.rpc, col, row, a.min_alt, lon, lat);
    utm_forward(a.utm, lat, lon, e1, n1);
    if (a.geo) { a.geo[i * 8 + 4] = lon; a.geo[i * 8 + 5] = lat; a.geo[i * 8 + 6] = e1; a.geo[i * 8 + 7] = n1; }
    const double dx = e1 - e0, dy = n1 - n0, dz = a.min_alt - a.max_alt;
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    // the reference stores these eight numbers as float32 (datasets/satellite.py:119-120) before normalising
    const float raw[8] = {(float)e0, (float)n0, (float)a.max_alt, (float)(dx / len), (float)(dy / len), (float)(dz / len), 0.f, (float)len};
    if (a.raw8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a.raw8[i * 8 + k] = raw[k];
    }
    if (!a.rays) return;
    // normalize_rays (:124-139) in fp64 on the fp32 values, offsets/scales being fp32 as the dataset keeps them
    double on[3], en[3], d[3], nrm = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double o = raw[k], ee = (double)raw[k] + (double)raw[3 + k] * (double)raw[7];
        on[k] = (o - (double)a.offset[k]) / (double)a.scale[k];
        en[k] = (ee - (double)a.offset[k]) / (double)a.scale[k];
        d[k] = en[k] - on[k];
        nrm += d[k] * d[k];
    }
    nrm = sqrt(nrm);
    double sun[3], sn = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { sun[k] = a.sun[k] / (double)a.scale[k]; sn += sun[k] * sun[k]; }
    sn = sqrt(sn);
    float* o = a.rays + i * 11;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = (float)on[k]; o[3 + k] = (float)(d[k] / nrm); o[8 + k] = (float)(sun[k] / sn); }
    o[6] = 0.f;
    o[7] = (float)nrm;
}

}  // namespace

hipError_t eo_launch_raygen(const RayGenArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_raygen, dim3((unsigned)((a.n + 127) / 128)), dim3(128), 0, st, a);
    return hipGetLastError();
}
