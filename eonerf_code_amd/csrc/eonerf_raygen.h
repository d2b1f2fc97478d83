// Argument block of the RPC ray-generation kernel (eonerf_raygen.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RpcModel {       // rpcm "rpcm" dict format, image = f(ground)
    double col_num[20], col_den[20], row_num[20], row_den[20];
    double row_offset, col_offset, lat_offset, lon_offset, alt_offset;
    double row_scale, col_scale, lat_scale, lon_scale, alt_scale;
};
struct UtmParams { double lon0_deg, e, k0A, false_north, alpha[6]; };
struct RayGenArgs {
    RpcModel rpc; UtmParams utm;
    const double *cols, *rows;      // explicit pixels, or nullptr: full width x height grid, row-major
    long n; int width;
    double min_alt, max_alt;
    double sun[3];                  // get_dir_vec_from_el_az(90 - sun_elevation, sun_azimuth), fp64
    float offset[3], scale[3];      // scene.loc_utm X/Y/Z offset and scale (fp32, datasets/satellite.py:303-307)
    float* raw8;                    // optional [n,8] un-normalised rays (the reference's <cache_dir>/<img>.data payload)
    float* rays;                    // optional [n,11] normalised rays
    double* geo;                    // optional [n,8] fp64: lon, lat (deg), east, north (m) at max_alt, then the same at min_alt
};
hipError_t eo_launch_raygen(const RayGenArgs& a, hipStream_t st);
