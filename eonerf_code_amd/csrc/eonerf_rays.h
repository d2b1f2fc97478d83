// Argument blocks of the per-ray kernels (eonerf_rays.hip, eonerf_rays_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// per-ray record produced by compositing (floats)
constexpr int RR_DEPTH = 0, RR_ALB = 1, RR_TS = 4, RR_TB = 5, RR_WSUM = 6, RR_AMB = 7, RR_GEO = 10, RAY_REC = 12;

struct AmbientW { const float *w1, *b1, *w2, *b2; };   // ambient_mlp: [128][27], [128], [3][128], [3]  (fp32)

struct SampleArgs {
    const float* rays;        // [R][11] fp32: o3 d3 near far sun3  (datasets/satellite.py:23-26)
    const int64_t* img_idx;   // [R] or nullptr
    const float* zsteps;      // [128] = torch.linspace(0,1,128)
    const float* u;           // [R][128] jitter of this pass
    const float* u_retry;     // [R][128] or nullptr: noise of the "some ray is empty -> resample" branch
    const float* depth;       // sun pass: rendered depth per ray
    int depth_stride;
    int n_rays;
    int sun_pass;             // 1: origin = o + depth*d, dir = -sun, near = 0
    int patch_last;           // 1: last interval of each ray ends at 1e10 (camera pass)
    int *cnt_first, *cnt_retry, *counts, *offsets;   // [R], [R], [R], [R+1]
    int* flags;               // bit0: retry taken
    int* n_pts;
    float *px, *py, *pz, *tmid, *delta;   // [p_pad] compact outputs
    int* simg;
};

struct CompositeArgs {
    const float* rays;
    const int *offsets, *counts;
    const float *sigma, *delta, *tmid, *albedo, *ts, *tb;
    int p_pad, n_rays;
    int shadow_only, depth_only;
    AmbientW amb;
    float* ray_out;           // [R][RAY_REC]
};

struct ShadeArgs {
    const float* ray_rec;
    const int64_t* img_idx;
    const float* radiometric;   // [n_img][9] or nullptr
    const int *pts_first, *sc_counts;
    int n_rays, use_shadow, eval;
    float* out;                 // [R][21]
};

hipError_t eo_launch_sampler(const SampleArgs& a, hipStream_t st);
hipError_t eo_launch_composite_fwd(const CompositeArgs& a, hipStream_t st);
hipError_t eo_launch_shade_fwd(const ShadeArgs& a, hipStream_t st);
hipError_t eo_launch_points_to_soa(const float* xyz, const int64_t* img, int n, int p_pad, float* px, float* py, float* pz,
                                   int* simg, int* n_pts, hipStream_t st);
hipError_t eo_launch_ambient_points(const AmbientW& w, const float* sun, int n, float* out, hipStream_t st);
hipError_t eo_launch_soa3_to_aos(const float* soa, int p_pad, int n, float* aos, hipStream_t st);
