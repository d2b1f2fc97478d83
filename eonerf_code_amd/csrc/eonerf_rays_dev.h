// Device helpers shared by the per-ray forward and backward kernels.
#pragma once
#include "eonerf_common.h"
#include "eonerf_rays.h"
#include <type_traits>

namespace {

constexpr int RAYS_PER_BLOCK = 4;   // 256 threads

// ---- wave helpers -------------------------------------------------------------------------------------------
EO_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes
EO_DEV float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
// inclusive SUFFIX scan (sum over lanes >= lane)
EO_DEV float wave_suffix_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    return v;
}

// ---- ambient head for one ray, computed by the whole wave (radiance_fields/eonerf.py:132-139,163-164) ------
struct AmbientRay { float out[3]; float hid[2]; float pre[3]; float enc[27]; };

EO_DEV void sun_encoding(float sx, float sy, float sz, float* enc) {   // mlp.py:190-208, L=4, fp32
    enc[0] = sx; enc[1] = sy; enc[2] = sz;
    const float c[3] = {sx, sy, sz};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xb = c[d] * (float)(1 << k);
            enc[3 + 3 * k + d] = sinf(xb);
            enc[15 + 3 * k + d] = sinf(xb + EO_PI_2_F);
        }
}

// one encoding element per lane (lanes 0..26), broadcast with shuffles: same arithmetic as sun_encoding
EO_DEV void sun_encoding_wave(float sx, float sy, float sz, int lane, float* enc) {
    float v = 0.f;
    if (lane < 27) {
        if (lane < 3) v = lane == 0 ? sx : (lane == 1 ? sy : sz);
        else {
            const int q = (lane - 3) % 12, k = q / 3, d = q % 3;
            const float c = d == 0 ? sx : (d == 1 ? sy : sz);
            const float xb = c * (float)(1 << k);
            v = lane < 15 ? sinf(xb) : sinf(xb + EO_PI_2_F);
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) enc[i] = __shfl(v, i, 64);
}

EO_DEV AmbientRay ambient_forward(const AmbientW& w, float sx, float sy, float sz, int lane) {
    AmbientRay r;
    sun_encoding_wave(sx, sy, sz, lane, r.enc);
    float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = lane + 64 * k;
        float acc = w.b1[j];
        for (int i = 0; i < 27; ++i) acc = fmaf(w.w1[j * 27 + i], r.enc[i], acc);
        r.hid[k] = fmaxf(acc, 0.f);
#pragma unroll
        for (int o = 0; o < 3; ++o) part[o] = fmaf(w.w2[o * 128 + j], r.hid[k], part[o]);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) { r.pre[o] = wave_sum(part[o]) + w.b2[o]; r.out[o] = sigmoid_f(r.pre[o]); }
    return r;
}

// ---- compositing forward ------------------------------------------------------------------------------------
//   sd = sigma*delta;  T = exp(-exclusive_sum(sd));  alpha = 1-exp(-sd);  w = T*alpha   (nerfacc v0.5.2 volrend,
//   call sites radiance_fields/eonerf.py:229-243)            per-ray sums of w*{mid, albedo, ts, tb, 1}
// SPL = samples per lane: a ray has n_samples - 1 = 64 SPL - 1 intervals at most (n_samples = int(2 / render_step_size) = 64, 128 or
// 256, sat_rendering.py:64), element i = lane + 64 k sits in slot k of its lane
template <int SPL> struct RayWeights { float w[SPL], T[SPL], sd[SPL]; float total; };

template <int SPL>
EO_DEV RayWeights<SPL> ray_weights(const float* sigma, const float* delta, int off, int n, int lane) {
    RayWeights<SPL> r;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        r.sd[k] = i < n ? sigma[off + i] * delta[off + i] : 0.f;
    }
    // exclusive prefix = inclusive prefix of the PREVIOUS lane (never "inclusive - self": the last interval has
    // sigma*delta ~ 1e10 and would cancel the whole prefix), plus the total of the 64-element groups in front
    float carry = 0.f, ex[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const float inc = wave_incl_scan(r.sd[k], lane);
        const float prev = __shfl_up(inc, 1, 64);
        ex[k] = lane == 0 ? carry : carry + prev;
        carry += __shfl(inc, 63, 64);
    }
    r.total = carry;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        r.T[k] = expf(-ex[k]);
        r.w[k] = i < n ? r.T[k] * (1.f - expf(-r.sd[k])) : 0.f;
    }
    return r;
}

// run f(std::integral_constant<int, SPL>) for the sample slots per lane n_samples needs (<= 64 -> 1, <= 128 -> 2, <= 256 -> 4)
template <class F> inline void eo_dispatch_spl(int n_samples, F&& f) {
    if (n_samples <= 64) f(std::integral_constant<int, 1>());
    else if (n_samples > 128) f(std::integral_constant<int, 4>());      // (129 .. 256: a three-slot instance would save a quarter of the per-ray work of kernels that are < 2 % of a step)
    else f(std::integral_constant<int, 2>());
}


}  // namespace
