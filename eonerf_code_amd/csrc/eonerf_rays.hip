// Per-ray kernels of the EO-NeRF hot path: stratified sampler + cube filter (H3), sample compaction, alpha
// compositing as a per-ray wavefront scan (H7), shadow-ray transmittance (H8), S-NeRF irradiance + radiometric
// affine + output packing (H9), and their backward passes.  One wave (64 lanes) owns one ray; a ray has at most
// n_samples - 1 intervals, n_samples = int(2 / render_step_size) <= 256 (sat_rendering.py:64) -> SPL = 1, 2 or 4 sample slots per lane
// (i = lane + 64 k; the kernels are instantiated per SPL; slots i >= n_samples - 1 hold no interval and are masked).
#include "eonerf_common.h"
#include "eonerf_rays.h"
#include "eonerf_rays_dev.h"

namespace {

// ---- the SatNeRF sampler for one ray (sat_rendering.py:46-84), evaluated without FMA contraction so that
//      t values and the cube-filter decisions are bit-identical to the reference's fp32 torch ops ------------
template <int SPL> struct RaySamples {
    float ts[SPL], te[SPL], mid[SPL], x[SPL], y[SPL], z[SPL];
    bool valid[SPL];
};

EO_DEV float zval(const float* zsteps, float near, int i) {
    // near * (1 - s) + (near + 2) * s      (sat_rendering.py:60-68)
    const float s = zsteps[i];
    return __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, s)), __fmul_rn(__fadd_rn(near, 2.0f), s));
}
// ns = n_samples = int(2 / render_step_size) (sat_rendering.py:64); zsteps = linspace(0, 1, ns)
EO_DEV float zperturbed(const float* zsteps, float near, int i, float u, int ns) {
    const float zi = zval(zsteps, near, i);
    const float lower = i == 0 ? zi : __fmul_rn(0.5f, __fadd_rn(zval(zsteps, near, i - 1), zi));
    const float upper = i == ns - 1 ? zi : __fmul_rn(0.5f, __fadd_rn(zi, zval(zsteps, near, i + 1)));
    return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));       // perturb_z_vals, :46-54
}

// ---- jitter source: caller-provided arrays (parity tests, torch.rand) or the in-kernel Philox4x32-10 stream -------------
//      Philox (Salmon et al. 2011, the generator behind torch.rand on GPUs): counter = (ray, lane, draw, call), key = seed;
//      one counter gives the lane's (up to four) jitters (samples lane + 64 k); 24 random bits -> [0, 1) fp32, as torch.rand.
EO_DEV void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
EO_DEV void philox_u4(uint64_t seed, uint32_t ray, uint32_t lane, uint32_t draw, uint32_t call, float (&u)[4]) {
    uint32_t c[4] = {ray, lane, draw, call};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = (float)(c[k] >> 8) * 0x1p-24f;
}
// jitters of samples lane + 64 k of draw `draw` (0 camera, 1 camera retry, 2 sun) of ray `ray`
template <int SPL>
EO_DEV void jitter(const SampleArgs& a, const float* u_arr, int draw, int ray, int lane, float (&u)[SPL]) {
    if (u_arr) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) u[k] = lane + 64 * k < a.n_samples ? u_arr[(size_t)ray * a.n_samples + lane + 64 * k] : 0.f;      // [R][n_samples]
    } else {
        float u4[4];
        philox_u4(a.seed, (uint32_t)ray, (uint32_t)lane, (uint32_t)draw, a.call, u4);
#pragma unroll
        for (int k = 0; k < SPL; ++k) u[k] = u4[k];
    }
}

template <int SPL>
EO_DEV RaySamples<SPL> sample_ray(const float* zsteps, int ns, bool perturb, const float (&u)[SPL], float near, float ox, float oy, float oz,
                                  float dx, float dy, float dz, int lane) {
    RaySamples<SPL> s;
    float zs[SPL], zn[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {      // perturb=False: :70-71 skipped.  Slots beyond the last z value repeat it (never used: masked below)
        const int i = lane + 64 * k < ns ? lane + 64 * k : ns - 1;
        zs[k] = perturb ? zperturbed(zsteps, near, i, u[k], ns) : zval(zsteps, near, i);
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        zn[k] = __shfl_down(zs[k], 1, 64);
        if (k + 1 < SPL) { const float z0 = __shfl(zs[k + 1 < SPL ? k + 1 : k], 0, 64); if (lane == 63) zn[k] = z0; }      // (last group, lane 63: interval NS - 1 does not exist)
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        s.ts[k] = zs[k];
        s.te[k] = __fadd_rn(zs[k], __fsub_rn(zn[k], zs[k]));               // a + (b - a), :74
        s.mid[k] = __fdiv_rn(__fadd_rn(s.ts[k], s.te[k]), 2.0f);           // :79
        s.x[k] = __fadd_rn(ox, __fmul_rn(dx, s.mid[k]));                    // :80
        s.y[k] = __fadd_rn(oy, __fmul_rn(dy, s.mid[k]));
        s.z[k] = __fadd_rn(oz, __fmul_rn(dz, s.mid[k]));
        const bool inside = fabsf(s.x[k]) < 1.0f && fabsf(s.y[k]) < 1.0f && fabsf(s.z[k]) < 1.0f;   // :18-22
        s.valid[k] = inside && lane + 64 * k < ns - 1;      // interval i = [z_i, z_{i+1}], i < n_samples - 1 (:74-76)
    }
    return s;
}
template <int SPL> EO_DEV int count_valid(const RaySamples<SPL>& s) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) n += __popcll(__ballot(s.valid[k]));
    return n;
}

struct RayGeom { float ox, oy, oz, dx, dy, dz, near; };

// camera rays come from the [R,11] table; sun rays start at the rendered surface point and look at the sun
// (sat_rendering.py:90-91: origin = o + depth*d, dir = -sundir, near = 0)
EO_DEV RayGeom sun_geom(const float* r, float depth) {
    RayGeom g;
    g.ox = __fadd_rn(r[0], __fmul_rn(depth, r[3]));
    g.oy = __fadd_rn(r[1], __fmul_rn(depth, r[4]));
    g.oz = __fadd_rn(r[2], __fmul_rn(depth, r[5]));
    g.dx = -r[8]; g.dy = -r[9]; g.dz = -r[10];
    g.near = 0.f;
    return g;
}
EO_DEV RayGeom ray_geom(const SampleArgs& a, int ray) {
    const float* r = a.rays + (size_t)ray * 11;
    RayGeom g;
    if (a.sun_pass) {
        g = sun_geom(r, a.depth[(size_t)ray * a.depth_stride]);
    } else {
        g.ox = r[0]; g.oy = r[1]; g.oz = r[2]; g.dx = r[3]; g.dy = r[4]; g.dz = r[5];
        g.near = r[6];
    }
    return g;
}

// ---- kernel 1: count samples per ray (for both the first draw and the "retry" draw; k_scan decides which one counts) ----
template <int SPL>
EO_DEV void count_ray(const SampleArgs& a, int ray, int lane, const RayGeom& g) {
    float u[SPL];
    jitter<SPL>(a, a.u, a.sun_pass ? 2 : 0, ray, lane, u);
    const RaySamples<SPL> s = sample_ray<SPL>(a.zsteps, a.n_samples, a.perturb, u, g.near, g.ox, g.oy, g.oz, g.dx, g.dy, g.dz, lane);
    const int cnt = count_valid(s);
    int cnt_retry = cnt;
    if (a.retry) {
        // the reference's retry passes near=None -> zeros (sat_rendering.py:262)
        jitter<SPL>(a, a.u_retry, 1, ray, lane, u);
        const RaySamples<SPL> s2 = sample_ray<SPL>(a.zsteps, a.n_samples, a.perturb, u, 0.f, g.ox, g.oy, g.oz, g.dx, g.dy, g.dz, lane);
        cnt_retry = count_valid(s2);
    }
    if (lane == 0) {
        a.cnt_first[ray] = cnt;
        a.cnt_retry[ray] = cnt_retry;
    }
}
template <int SPL>
__global__ __launch_bounds__(256) void k_count(SampleArgs a) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    count_ray<SPL>(a, ray, lane, ray_geom(a, ray));
}

// ---- kernel 2: exclusive scan of the chosen counts -> offsets[R+1]; n_pts; pts_per_ray (first draw) ---------
__global__ __launch_bounds__(1024) void k_scan(SampleArgs a) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // "resample if any ray is empty" (sat_rendering.py:260-262): decided here, from the counts of the first draw, and left in flags[0]
    // for k_emit (a plain store: nothing has to be zeroed before the sampler runs)
    bool retry = false;
    if (a.retry) {
        int any = 0;
        for (int i = tid; i < a.n_rays; i += 1024) any |= a.cnt_first[i] == 0 ? 1 : 0;
        retry = __syncthreads_or(any) != 0;
        if (tid == 0) a.flags[0] = retry ? 1 : 0;
    }
    const int* cnt = retry ? a.cnt_retry : a.cnt_first;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < a.n_rays; base += 1024) {
        const int i = base + tid;
        const int v = i < a.n_rays ? cnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int c = carry;
        if (i < a.n_rays) {
            a.offsets[i] = c + woff + incl - v;
            a.counts[i] = v;
        }
        __syncthreads();
        if (tid == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (tid == 0) { a.offsets[a.n_rays] = carry; *a.n_pts = carry; if (a.n_pts_copy) *a.n_pts_copy = carry; }
}

// ---- kernel 3: recompute the samples and write them compactly ---------------------------------------------
// SCAN: the launch has no k_scan in front (batches of up to SCAN_FUSED_MAX_RAYS rays): every block sums the counts of the rays in front of
// its own itself -- 2 x 16 KB of L2-resident reads per block at 4096 rays, a few microseconds for the whole grid, against a launch of
// its own for a 4096-element scan -- and decides the "resample if any ray is empty" branch (sat_rendering.py:260-262) from the same pass;
// block 0 leaves the totals (n_pts, offsets[R], the retry flag).
constexpr int SCAN_FUSED_MAX_RAYS = 8192;
template <int SPL, bool SCAN>
__global__ __launch_bounds__(256) void k_emit(SampleArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ray = blockIdx.x * RAYS_PER_BLOCK + wave;
    bool retry_scan = false;
    int off_scan = 0;
    if constexpr (SCAN) {
        __shared__ int s_red[RAYS_PER_BLOCK][5];
        const int base = blockIdx.x * RAYS_PER_BLOCK;
        int v[5] = {0, 0, 0, 0, 0};      // counts in front of this block (first draw, retry draw), totals (first, retry), any empty ray
        for (int i = threadIdx.x; i < a.n_rays; i += 256) {
            const int cf = a.cnt_first[i], cr = a.retry ? a.cnt_retry[i] : cf;
            v[2] += cf; v[3] += cr; v[4] |= cf == 0 ? 1 : 0;
            if (i < base) { v[0] += cf; v[1] += cr; }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v[q], o, 64); v[q] = q == 4 ? (v[q] | t) : (v[q] + t); }
        }
        if (lane == 0) { for (int q = 0; q < 5; ++q) s_red[wave][q] = v[q]; }
        __syncthreads();
        int t[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) t[q] = q == 4 ? (s_red[0][q] | s_red[1][q] | s_red[2][q] | s_red[3][q]) : (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
        retry_scan = a.retry && t[4] != 0;
        const int* cnt = retry_scan ? a.cnt_retry : a.cnt_first;
        off_scan = retry_scan ? t[1] : t[0];
        for (int j = base; j < ray && j < a.n_rays; ++j) off_scan += cnt[j];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int total = retry_scan ? t[3] : t[2];
            a.flags[0] = retry_scan ? 1 : 0;
            a.offsets[a.n_rays] = total; *a.n_pts = total;
            if (a.n_pts_copy) *a.n_pts_copy = total;
        }
        if (ray < a.n_rays && lane == 0) { a.offsets[ray] = off_scan; a.counts[ray] = cnt[ray]; }
    }
    if (ray >= a.n_rays) return;
    if (a.digest) {      // (eonerf_presample only) lanes 0..7: the seven table words the camera sampler reads and the image index
        unsigned long long d = 0ull;
        if (lane < 8) {
            const uint32_t w = lane < 7 ? __float_as_uint(a.rays[(size_t)ray * 11 + lane]) : (a.img_idx ? (uint32_t)a.img_idx[ray] : 0u);
            d = ray_word_digest(w, 8u * (uint32_t)ray + (uint32_t)lane);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane == 0) atomicAdd(a.digest, d);
    }
    const RayGeom g = ray_geom(a, ray);
    const bool retry = SCAN ? retry_scan : (a.retry && (*a.flags & 1));
    float u[SPL];
    jitter<SPL>(a, retry ? a.u_retry : a.u, retry ? 1 : (a.sun_pass ? 2 : 0), ray, lane, u);
    const RaySamples<SPL> s = sample_ray<SPL>(a.zsteps, a.n_samples, a.perturb, u, retry ? 0.f : g.near, g.ox, g.oy, g.oz, g.dx, g.dy, g.dz, lane);
    unsigned long long m[SPL];
    int before[SPL], n = 0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) { m[k] = __ballot(s.valid[k]); before[k] = n; n += __popcll(m[k]); }
    const int off = SCAN ? off_scan : a.offsets[ray];
    const unsigned long long below = (1ull << lane) - 1ull;
    const int img = a.img_idx ? (int)a.img_idx[ray] : 0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        if (!s.valid[k]) continue;
        const int rank = before[k] + __popcll(m[k] & below);
        const int p = off + rank;
        a.px[p] = s.x[k]; a.py[p] = s.y[k]; a.pz[p] = s.z[k];
        a.simg[p] = img;
        a.tmid[p] = s.mid[k];
        // camera pass: the last interval of every ray ends at 1e10 (radiance_fields/eonerf.py:218-220)
        const float te = (a.patch_last && rank == n - 1) ? 1e10f : s.te[k];
        a.delta[p] = __fsub_rn(te, s.ts[k]);
        if (a.o_ts) { a.o_ray[p] = ray; a.o_ts[p] = s.ts[k]; a.o_te[p] = s.te[k]; }
    }
}

// ---- shading + output packing (sat_rendering.py:265-312) of one ray; `r` = its complete ray record -----------------------
EO_DEV void shade_ray(const ShadeArgs& a, int ray, const float* r) {
    float* o = a.out + (size_t)ray * 21;
    const float wsum = r[RR_WSUM];
    const float geo = a.use_shadow ? r[RR_GEO] : 1.0f;
    const float ts = r[RR_TS];
    const float s = a.use_shadow ? geo * ts : 1.0f;                       // :269-276
    const long img = a.eval ? a.img_idx[0] : a.img_idx[ray];               // :288-291
    const float* T = a.radiometric ? a.radiometric + img * 9 : nullptr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float alb = r[RR_ALB + c];
        const float amb = (wsum * r[RR_AMB + c]) * 0.2f;                   // accumulate (eonerf.py:240) then *0.2 (:265)
        float rgb = alb * s + (1.f - s) * (amb * alb);                     // :294
        const float A = T ? T[c] : 1.f, b = T ? T[3 + c] : 0.f;
        rgb = A * rgb + b;
        o[c] = fminf(fmaxf(rgb, 0.f), 1.f);                                // :304-305
        o[4 + c] = alb;
        o[7 + c] = amb;
        o[18 + c] = A * alb + b;                                           // :306
    }
    o[3] = r[RR_DEPTH];
    o[10] = geo;
    o[11] = ts;
    o[12] = r[RR_TB];
    o[13] = 1.0f;                                                          // entropy, eonerf.py:246
    o[14] = (float)a.pts_first[ray];                                       // counts of the FIRST draw (:259, stale on retry)
    o[15] = a.use_shadow ? (float)a.sc_counts[ray] : 1.0f;                 // :272 / :96
    o[16] = 1.0f; o[17] = 1.0f;                                            // opacity_after_surface, :283
}
__global__ __launch_bounds__(256) void k_shade_fwd(ShadeArgs a) {
    const int ray = blockIdx.x * 256 + threadIdx.x;
    if (ray >= a.n_rays) return;
    shade_ray(a, ray, a.ray_rec + (size_t)ray * RAY_REC);
}

template <int SPL>
__global__ __launch_bounds__(256) void k_composite_fwd(CompositeArgs a) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    const int off = a.offsets[ray], n = a.counts[ray];
    const RayWeights<SPL> rw = ray_weights<SPL>(a.sigma, a.delta, off, n, lane);
    if (a.shadow_only) {
        // geo_shadow = T at the LAST valid sample (exclusive), 1 for an empty ray (sat_rendering.py:112-116)
        const int last = n - 1;
        float Tsel = rw.T[0];
#pragma unroll
        for (int k = 1; k < SPL; ++k) Tsel = (last >> 6) == k ? rw.T[k] : Tsel;      // (wave-uniform selection)
        const float Tl = __shfl(Tsel, last & 63, 64);
        if (lane == 0) {
            float* o = a.ray_out + (size_t)ray * RAY_REC;
            o[RR_GEO] = n > 0 ? Tl : 1.0f;
            if (a.do_shade) {      // the record is complete: the camera pass' fields were written by an earlier launch
                float rec[RAY_REC];
#pragma unroll
                for (int i = 0; i < RAY_REC; ++i) rec[i] = o[i];
                rec[RR_GEO] = n > 0 ? Tl : 1.0f;
                shade_ray(a.shade, ray, rec);
            }
        }
        return;
    }
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // depth, albedo3, ts, tb, wsum
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        if (i < n) {
            const int p = off + i;
            const float w = rw.w[k];
            acc[0] += w * a.tmid[p];
            if (!a.depth_only) {
                acc[1] += w * a.albedo[p];
                acc[2] += w * a.albedo[(size_t)a.p_pad + p];
                acc[3] += w * a.albedo[2 * (size_t)a.p_pad + p];
                acc[4] += w * a.ts[p];
                acc[5] += w * a.tb[p];
            }
            acc[6] += w;
        }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = wave_sum(acc[j]);
    float amb[3] = {0.f, 0.f, 0.f};
    if (!a.depth_only) {
        const float* r = a.rays + (size_t)ray * 11;
        const AmbientRay ar = ambient_forward(a.amb, r[8], r[9], r[10], lane);
        amb[0] = ar.out[0]; amb[1] = ar.out[1]; amb[2] = ar.out[2];
        if (a.amb_save) {
            float* sv = a.amb_save + (size_t)ray * 160;
            float ev = 0.f;
#pragma unroll
            for (int i = 0; i < 27; ++i) ev = lane == i ? ar.enc[i] : ev;
            if (lane < 27) sv[lane] = ev;
            sv[32 + lane] = ar.hid[0];
            sv[96 + lane] = ar.hid[1];
        }
    }
    if (lane == 0) {
        float* o = a.ray_out + (size_t)ray * RAY_REC;
        o[RR_DEPTH] = acc[0];
        o[RR_ALB + 0] = acc[1]; o[RR_ALB + 1] = acc[2]; o[RR_ALB + 2] = acc[3];
        o[RR_TS] = acc[4];
        o[RR_TB] = acc[5] + 0.05f;                      // beta_min, eonerf.py:87,243
        o[RR_WSUM] = acc[6];
        o[RR_AMB + 0] = amb[0]; o[RR_AMB + 1] = amb[1]; o[RR_AMB + 2] = amb[2];   // sigmoid output of the head
        o[RR_GEO] = 1.0f;
        if (a.do_shade) {
            float rec[RAY_REC] = {};
            rec[RR_DEPTH] = acc[0]; rec[RR_ALB] = acc[1]; rec[RR_ALB + 1] = acc[2]; rec[RR_ALB + 2] = acc[3]; rec[RR_TS] = acc[4];
            rec[RR_TB] = acc[5] + 0.05f; rec[RR_WSUM] = acc[6]; rec[RR_AMB] = amb[0]; rec[RR_AMB + 1] = amb[1]; rec[RR_AMB + 2] = amb[2];
            rec[RR_GEO] = 1.0f;
            shade_ray(a.shade, ray, rec);
        }
    }
    // the shadow ray of this ray starts at the surface point it has just rendered: count its samples here (acc[0] = depth on every lane)
    if (a.count_sun) count_ray<SPL>(a.sun, ray, lane, sun_geom(a.rays + (size_t)ray * 11, acc[0]));
}

// ---- caller-provided flattened samples (radiance_fields/eonerf.py:196-220: gather, mid points, last t_end := 1e10) ----
__global__ void k_packed_bounds(PackedArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    const int64_t ray = a.ray_indices[p];
    if (p == 0 || a.ray_indices[p - 1] != ray) a.offsets[ray] = p;
    if (p == a.n - 1 || a.ray_indices[p + 1] != ray) a.counts[ray] = p + 1;       // end index, turned into a count below
}
__global__ void k_packed_emit(PackedArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) { a.offsets[a.n_rays] = a.n; *a.n_pts = a.n; }
    if (p >= a.n) return;
    const int64_t ray = a.ray_indices[p];
    const float* r = a.rays + (size_t)ray * 11;
    const float ts = a.t_starts[p], te = a.t_ends[p];
    const float mid = __fdiv_rn(__fadd_rn(ts, te), 2.0f);
    a.px[p] = __fadd_rn(r[0], __fmul_rn(r[3], mid));
    a.py[p] = __fadd_rn(r[1], __fmul_rn(r[4], mid));
    a.pz[p] = __fadd_rn(r[2], __fmul_rn(r[5], mid));
    a.tmid[p] = mid;
    const bool last = p == a.n - 1 || a.ray_indices[p + 1] != ray;
    a.delta[p] = __fsub_rn(last ? 1e10f : te, ts);
    a.simg[p] = a.img_idx ? (int)a.img_idx[ray] : 0;
}
__global__ void k_packed_counts(PackedArgs a) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= a.n_rays) return;
    const int end = a.counts[ray];
    a.counts[ray] = end > 0 ? end - a.offsets[ray] : 0;
}
__global__ void k_int_to_float(const int* src, int n, float* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
__global__ void k_rendering_out(RenderingOutArgs a) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= a.n_rays) return;
    const float* r = a.ray_rec + (size_t)ray * RAY_REC;
    a.depth[ray] = r[RR_DEPTH];
    if (!a.albedo) return;
    for (int c = 0; c < 3; ++c) { a.albedo[3 * ray + c] = r[RR_ALB + c]; a.ambient[3 * ray + c] = r[RR_WSUM] * r[RR_AMB + c]; }
    a.beta[ray] = r[RR_TB]; a.ts[ray] = r[RR_TS]; a.entropy[ray] = 1.0f;
}

// ---- small utility kernels ---------------------------------------------------------------------------------
__global__ void k_points_to_soa(const float* xyz, const int64_t* img, int n, int p_pad, float* px, float* py, float* pz,
                                int* simg, int* n_pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_pts = n;
    if (i >= p_pad) return;
    if (i < n) {
        px[i] = xyz[3 * (size_t)i]; py[i] = xyz[3 * (size_t)i + 1]; pz[i] = xyz[3 * (size_t)i + 2];
        simg[i] = img ? (int)img[i] : 0;
    } else if (i < ((n + 255) & ~255)) {
        px[i] = 0.f; py[i] = 0.f; pz[i] = 0.f; simg[i] = 0;
    }
}

__global__ __launch_bounds__(256) void k_ambient_points(AmbientW w, const float* sun, int n, float* out) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const AmbientRay ar = ambient_forward(w, sun[3 * (size_t)i], sun[3 * (size_t)i + 1], sun[3 * (size_t)i + 2], lane);
    if (lane < 3) out[3 * (size_t)i + lane] = lane == 0 ? ar.out[0] : (lane == 1 ? ar.out[1] : ar.out[2]);
}

__global__ void k_soa3_to_aos(const float* soa, int p_pad, int n, float* aos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { aos[3 * (size_t)i] = soa[i]; aos[3 * (size_t)i + 1] = soa[(size_t)p_pad + i]; aos[3 * (size_t)i + 2] = soa[2 * (size_t)p_pad + i]; }
}

}  // namespace

hipError_t eo_launch_sampler(const SampleArgs& a, hipStream_t st, bool counted) {
    const int blocks = (a.n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
    eo_dispatch_spl(a.n_samples, [&](auto spl) {
        constexpr int SPL = decltype(spl)::value;
        if (!counted) hipLaunchKernelGGL(k_count<SPL>, dim3(blocks), dim3(256), 0, st, a);
        if (a.n_rays <= SCAN_FUSED_MAX_RAYS) {      // the scan rides in the emit kernel
            hipLaunchKernelGGL((k_emit<SPL, true>), dim3(blocks), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, a);
            hipLaunchKernelGGL((k_emit<SPL, false>), dim3(blocks), dim3(256), 0, st, a);
        }
    });
    return hipGetLastError();
}
hipError_t eo_launch_from_packed(const PackedArgs& a, hipStream_t st) {
    hipError_t e = hipMemsetAsync(a.counts, 0, sizeof(int) * a.n_rays, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.offsets, 0, sizeof(int) * (a.n_rays + 1), st);
    if (e != hipSuccess) return e;
    const int nb = (a.n + 255) / 256 > 0 ? (a.n + 255) / 256 : 1;
    hipLaunchKernelGGL(k_packed_bounds, dim3(nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_packed_counts, dim3((a.n_rays + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_packed_emit, dim3(nb), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_int_to_float(const int* src, int n, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_int_to_float, dim3((n + 255) / 256), dim3(256), 0, st, src, n, dst);
    return hipGetLastError();
}
hipError_t eo_launch_rendering_out(const RenderingOutArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_rendering_out, dim3((a.n_rays + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_composite_fwd(const CompositeArgs& a, hipStream_t st) {
    eo_dispatch_spl(a.n_samples, [&](auto spl) {
        hipLaunchKernelGGL(k_composite_fwd<decltype(spl)::value>, dim3((a.n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), dim3(256), 0, st, a);
    });
    return hipGetLastError();
}
hipError_t eo_launch_shade_fwd(const ShadeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_shade_fwd, dim3((a.n_rays + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_points_to_soa(const float* xyz, const int64_t* img, int n, int p_pad, float* px, float* py, float* pz,
                                   int* simg, int* n_pts, hipStream_t st) {
    hipLaunchKernelGGL(k_points_to_soa, dim3((p_pad + 255) / 256), dim3(256), 0, st, xyz, img, n, p_pad, px, py, pz, simg, n_pts);
    return hipGetLastError();
}
hipError_t eo_launch_ambient_points(const AmbientW& w, const float* sun, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_ambient_points, dim3((n + 3) / 4), dim3(256), 0, st, w, sun, n, out);
    return hipGetLastError();
}
hipError_t eo_launch_soa3_to_aos(const float* soa, int p_pad, int n, float* aos, hipStream_t st) {
    hipLaunchKernelGGL(k_soa3_to_aos, dim3((n + 255) / 256), dim3(256), 0, st, soa, p_pad, n, aos);
    return hipGetLastError();
}
