// Backward of the per-ray kernels (H10 of SURVEY.md 8a: autograd of sat_rendering.py:264-306 and of the nerfacc
// compositing calls at radiance_fields/eonerf.py:229-243, sat_rendering.py:106-116), plus the fused Adam update.
#include <stdlib.h>
#include <string.h>
#include "eonerf_common.h"
#include "eonerf_rays.h"
#include "eonerf_rays_dev.h"
#include "eonerf_ambient_dev.h"

namespace {

// ---- shading backward: d out[R,21] -> d ray record, d radiometric table ------------------------------------
__global__ __launch_bounds__(256) void k_shade_bwd(ShadeBwdArgs a) {
    extern __shared__ float s_dT[];                    // [n_img][6] block-local radiometric gradient (0 floats if unused)
    const int ray = blockIdx.x * 256 + threadIdx.x;
    if (a.zero_base) {
        u32x4* z = reinterpret_cast<u32x4*>(a.zero_base);
        const size_t n16 = a.zero_bytes / 16;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    const bool lds_acc = a.d_radiometric && a.lds_images > 0 && !a.d_rad_rays;
    if (lds_acc) {
        for (int i = threadIdx.x; i < a.lds_images * 6; i += 256) s_dT[i] = 0.f;
        __syncthreads();
    }
    if (a.chk_sum) {      // digest of the rays as they are now (see ray_word_digest): one atomic per wave
        unsigned long long d = 0ull;
        if (ray < a.n_rays) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t w = q < 7 ? __float_as_uint(a.chk_rays[(size_t)ray * 11 + q]) : (uint32_t)a.img_idx[ray];
                d += ray_word_digest(w, 8u * (uint32_t)ray + (uint32_t)q);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.chk_sum, d);
    }
    float loss_part = 0.f;
    if (ray < a.n_rays) {
        const float* r = a.ray_rec + (size_t)ray * RAY_REC;
        float go[21];
        if (a.loss_kind < 0) {
#pragma unroll
            for (int c = 0; c < 21; ++c) go[c] = a.d_out[(size_t)ray * 21 + c];
        } else {      // the loss and its gradient, exactly as k_loss computes them
#pragma unroll
            for (int c = 0; c < 21; ++c) go[c] = 0.f;
            const float* o = a.loss_out + (size_t)ray * 21;
            const int n = a.n_rays;
            const float inv = 1.f / (3.f * n);
            if (a.loss_kind == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { const float df = o[c] - a.loss_gt[(size_t)ray * 3 + c]; loss_part += df * df * inv; go[c] = 2.f * df * inv; }
            } else {
                const float beta = o[12], ib2 = 1.f / (beta * beta);
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) { const float df = o[c] - a.loss_gt[(size_t)ray * 3 + c]; sq += df * df; go[c] = df * ib2 * inv; }
                loss_part += 0.5f * sq * ib2 * inv + 0.5f * logf(beta) / n;
                go[12] = -sq * ib2 / beta * inv + 0.5f / (n * beta);
            }
        }
        float* g = a.g_ray + (size_t)ray * RAY_REC;
        const float wsum = r[RR_WSUM], ts = r[RR_TS];
        const float geo = a.use_shadow ? r[RR_GEO] : 1.0f;
        const float s = a.use_shadow ? geo * ts : 1.0f;
        const long img = a.eval ? a.img_idx[0] : a.img_idx[ray];
        const float* T = a.radiometric ? a.radiometric + img * 9 : nullptr;
        float g_s = 0.f, g_wsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float alb = r[RR_ALB + c], head = r[RR_AMB + c];
            const float amb = (wsum * head) * 0.2f;
            const float pre = alb * s + (1.f - s) * (amb * alb);
            const float A = T ? T[c] : 1.f, b = T ? T[3 + c] : 0.f;
            const float lin = A * pre + b;
            const float g_lin = (lin >= 0.f && lin <= 1.f) ? go[c] : 0.f;        // torch.clip backward (inclusive bounds)
            const float g_shl = go[18 + c];
            if (a.d_rad_rays) {
                a.d_rad_rays[(size_t)ray * 6 + c] = g_lin * pre + g_shl * alb;
                a.d_rad_rays[(size_t)ray * 6 + 3 + c] = g_lin + g_shl;
            } else if (a.d_radiometric) {
                const float dA = g_lin * pre + g_shl * alb, db = g_lin + g_shl;
                if (lds_acc) { atomicAdd(&s_dT[img * 6 + c], dA); atomicAdd(&s_dT[img * 6 + 3 + c], db); }
                else { atomicAdd(a.d_radiometric + img * 9 + c, dA); atomicAdd(a.d_radiometric + img * 9 + 3 + c, db); }
            }
            const float g_pre = g_lin * A;
            g[RR_ALB + c] = g_pre * (s + (1.f - s) * amb) + go[4 + c] + g_shl * A;
            const float g_amb = g_pre * (1.f - s) * alb + go[7 + c];
            g_s += g_pre * (alb - amb * alb);
            g_wsum += g_amb * 0.2f * head;
            g[RR_AMB + c] = g_amb * 0.2f * wsum;
        }
        g[RR_DEPTH] = go[3];
        g[RR_TS] = (a.use_shadow ? g_s * geo : 0.f) + go[11];
        g[RR_GEO] = a.use_shadow ? g_s * ts + go[10] : 0.f;
        g[RR_TB] = go[12];
        g[RR_WSUM] = g_wsum;
    }
    if (lds_acc) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.lds_images * 6; i += 256) {
            const float v = s_dT[i];
            if (v != 0.f) atomicAdd(a.d_radiometric + (i / 6) * 9 + (i % 6), v);
        }
    }
    if (a.loss_kind >= 0) {      // the scalar: per-block partials, summed in block order by the last block to arrive (as k_loss)
        __shared__ float s_part[4];
        __shared__ int s_last;
        loss_part = wave_sum(loss_part);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = loss_part;
        __syncthreads();
        if (threadIdx.x == 0) {
            a.loss_scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            __threadfence();
            int* ticket = reinterpret_cast<int*>(a.loss_scratch + LOSS_MAX_BLOCKS);
            s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (!s_last || threadIdx.x != 0) return;
        __threadfence();
        float total = a.loss_kind == 1 ? 1.5f : 0.f;                 // the constant 3/2 of the beta term (metrics.py:20)
        for (unsigned b = 0; b < gridDim.x; ++b) total += __hip_atomic_load(a.loss_scratch + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *a.loss = total;
        *reinterpret_cast<int*>(a.loss_scratch + LOSS_MAX_BLOCKS) = 0;
    }
}

// ---- shadow-ray transmittance backward: geo = exp(-sum_{j<last} sigma_j delta_j) ------------------------------
template <int SPL>
__global__ __launch_bounds__(256) void k_sun_composite_bwd(CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    const int off = a.offsets[ray], n = a.counts[ray];
    const float g_geo = a.g_ray[(size_t)ray * RAY_REC + RR_GEO];
    const float geo = a.ray_rec[(size_t)ray * RAY_REC + RR_GEO];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        if (i < n) a.g_sigma[off + i] = i < n - 1 ? -g_geo * geo * a.delta[off + i] : 0.f;
    }
}

// ---- camera compositing backward ---------------------------------------------------------------------------
//   w_i = T_i (1 - e_i), e_i = exp(-sd_i), T_i = exp(-sum_{j<i} sd_j)
//   dL/dsd_i = g_w_i T_i e_i - sum_{j>i} g_w_j w_j          dL/dsigma_i = delta_i dL/dsd_i
template <int SPL>
__global__ __launch_bounds__(256) void k_cam_composite_bwd(CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63, ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    // presampled forward: the rays the sampler read are not the rays of this backward -> sticky status bit (k_adam skips the update)
    if (a.chk_a && blockIdx.x == 0 && threadIdx.x == 0 && *a.chk_a != *a.chk_b) atomicOr(a.chk_status, EO_STATUS_PRESAMPLE_STALE);
    if (ray >= a.n_rays) return;
    const int off = a.offsets[ray], n = a.counts[ray];
    const float* g = a.g_ray + (size_t)ray * RAY_REC;
    float g_depth = g[RR_DEPTH];
    const float g_ts = g[RR_TS], g_tb = g[RR_TB], g_wsum = g[RR_WSUM];
    if (a.sun_g_pos) {      // d depth += sum over the ray's shadow samples of <d pos, viewdir>  (origin = o + depth*d, sat_rendering.py:90)
        const int so = a.sun_offsets[ray], sn = a.sun_counts[ray];
        const float* r = a.rays + (size_t)ray * 11;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int i = lane + 64 * k;
            if (i < sn) {
                const int p = so + i;
                acc += a.sun_g_pos[p] * r[3] + a.sun_g_pos[(size_t)a.p_pad + p] * r[4] + a.sun_g_pos[2 * (size_t)a.p_pad + p] * r[5];
            }
        }
        g_depth += wave_sum(acc);
    }
    const float g_alb[3] = {g[RR_ALB], g[RR_ALB + 1], g[RR_ALB + 2]};
    const RayWeights<SPL> rw = ray_weights<SPL>(a.sigma, a.delta, off, n, lane);
    float gw_w[SPL], gw[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        gw[k] = 0.f; gw_w[k] = 0.f;
        if (i < n) {
            const int p = off + i;
            const float w = rw.w[k];
            if (a.depth_only) {      // render_depth (radiance_fields/eonerf.py:172-194): depth is the only output
                gw[k] = g_depth * a.tmid[p];
                gw_w[k] = gw[k] * w;
                continue;
            }
            const float alb0 = a.albedo[p], alb1 = a.albedo[(size_t)a.p_pad + p], alb2 = a.albedo[2 * (size_t)a.p_pad + p];
            gw[k] = g_depth * a.tmid[p] + g_alb[0] * alb0 + g_alb[1] * alb1 + g_alb[2] * alb2 + g_ts * a.ts[p] + g_tb * a.tb[p] + g_wsum;
            gw_w[k] = gw[k] * w;
            a.g_albedo[p] = w * g_alb[0];
            a.g_albedo[(size_t)a.p_pad + p] = w * g_alb[1];
            a.g_albedo[2 * (size_t)a.p_pad + p] = w * g_alb[2];
            a.g_ts[p] = w * g_ts;
            a.g_tb[p] = w * g_tb;
        }
    }
    // exclusive suffix sums of g_w*w over the ray (element i = lane + 64k)
    float after[SPL], carry = 0.f;      // carry = the sum over the 64-element groups behind group k
#pragma unroll
    for (int k = SPL - 1; k >= 0; --k) {
        const float suf = wave_suffix_scan(gw_w[k], lane);
        const float nxt = __shfl_down(suf, 1, 64);
        after[k] = k == SPL - 1 ? (lane == 63 ? 0.f : nxt) : (lane == 63 ? 0.f : nxt) + carry;
        carry += __shfl(suf, 0, 64);
    }
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int i = lane + 64 * k;
        if (i < n) {
            const int p = off + i;
            const float e = expf(-rw.sd[k]);
            a.g_sigma[p] = a.delta[p] * (gw[k] * rw.T[k] * e - after[k]);
        }
    }
}

// ---- EONerfMLP.rendering's outputs backwards (radiance_fields/eonerf.py:229-247; forward: k_rendering_out):
//      ambient_rgb = wsum * head,  beta = sum w tb + beta_min,  entropy = ones (no gradient) ----
__global__ void k_rendering_out_bwd(RenderingOutBwdArgs a) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= a.n_rays) return;
    const float* r = a.ray_rec + (size_t)ray * RAY_REC;
    float* g = a.g_ray + (size_t)ray * RAY_REC;
    float g_wsum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ga = a.g_ambient ? a.g_ambient[3 * (size_t)ray + c] : 0.f;
        g[RR_ALB + c] = a.g_albedo ? a.g_albedo[3 * (size_t)ray + c] : 0.f;
        g[RR_AMB + c] = ga * r[RR_WSUM];
        g_wsum += ga * r[RR_AMB + c];
    }
    g[RR_DEPTH] = a.g_depth ? a.g_depth[ray] : 0.f;
    g[RR_TS] = a.g_ts ? a.g_ts[ray] : 0.f;
    g[RR_TB] = a.g_beta ? a.g_beta[ray] : 0.f;
    g[RR_WSUM] = g_wsum;
    g[RR_GEO] = 0.f;
    g[RR_GEO + 1] = 0.f;
}

__global__ __launch_bounds__(128 * AMB_STREAMS) void k_ambient_bwd(AmbientBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[AMB_LDS_F];
    ambient_bwd_body(a, blockIdx.x, gridDim.x, lds);
}

// ---- differentiable EONerfMLP.forward on caller-provided points (radiance_fields/eonerf.py:154-170): glue around the chains ----
// upstream gradients of the forward outputs (row-major, as autograd hands them over; any may be null = zero) -> the SoA
// arrays the backward chain reads
__global__ void k_field_grads_to_soa(const float* g_sigma, const float* g_albedo, const float* g_ts, const float* g_tb, int n, int p_pad,
                                     float* o_sigma, float* o_albedo, float* o_ts, float* o_tb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_sigma[i] = g_sigma ? g_sigma[i] : 0.f;
    if (o_albedo) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o_albedo[(size_t)c * p_pad + i] = g_albedo ? g_albedo[3 * (size_t)i + c] : 0.f;
        o_ts[i] = g_ts ? g_ts[i] : 0.f;
        o_tb[i] = g_tb ? g_tb[i] : 0.f;
    }
}
// transient embedding gradient, one point per thread (nn.Embedding backward = index_add)
__global__ void k_emb_grad_points(const float* g_emb, const int* simg, int n, float* d_emb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(g_emb + 4 * (size_t)i);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (v[e] != 0.f) atomicAdd(d_emb + 4 * (size_t)simg[i] + e, v[e]);
}
// ambient head backward per POINT (the render path evaluates the head once per ray, k_ambient_bwd): a wave per point recomputes
// the head (27 -> 128 ReLU -> 3 Sigmoid), lanes own hidden units j = lane, lane + 64; sums stay in registers across the wave's
// points and meet in one set of atomics per wave
__global__ __launch_bounds__(256) void k_ambient_points_bwd(AmbientW w, const float* sun, const float* g_amb, int n,
                                                             float* d_w1, float* d_b1, float* d_w2, float* d_b2) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    float dw1[2][27], db1[2] = {0.f, 0.f}, dw2[2][3], db2[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 27; ++i) dw1[k][i] = 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) dw2[k][o] = 0.f;
    }
    for (int i = wave; i < n; i += n_waves) {
        const AmbientRay ar = ambient_forward(w, sun[3 * (size_t)i], sun[3 * (size_t)i + 1], sun[3 * (size_t)i + 2], lane);
        float gpre[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) { gpre[o] = g_amb[3 * (size_t)i + o] * ar.out[o] * (1.f - ar.out[o]); db2[o] += gpre[o]; }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = lane + 64 * k;
            float ghid = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) { dw2[k][o] += gpre[o] * ar.hid[k]; ghid += w.w2[o * 128 + j] * gpre[o]; }
            if (ar.hid[k] <= 0.f) ghid = 0.f;
            db1[k] += ghid;
#pragma unroll
            for (int q = 0; q < 27; ++q) dw1[k][q] += ghid * ar.enc[q];
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = lane + 64 * k;
#pragma unroll
        for (int q = 0; q < 27; ++q) if (dw1[k][q] != 0.f) atomicAdd(d_w1 + j * 27 + q, dw1[k][q]);
        if (db1[k] != 0.f) atomicAdd(d_b1 + j, db1[k]);
#pragma unroll
        for (int o = 0; o < 3; ++o) if (dw2[k][o] != 0.f) atomicAdd(d_w2 + o * 128 + j, dw2[k][o]);
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o) if (db2[o] != 0.f) atomicAdd(d_b2 + o, db2[o]);
    }
}

// ---- deterministic mode: table[idx[r]][w] += contrib[r][w], one thread per (table row, w), rays in order ----
__global__ void k_table_reduce(const float* contrib, const int64_t* idx, int n_rays, int width, int stride, int n_rows, int eval_first, float* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * width) return;
    const int row = t / width, w = t % width;
    float acc = 0.f;
    const long first = eval_first ? idx[0] : 0;
    for (int r = 0; r < n_rays; ++r) {
        const long i = eval_first ? first : idx[r];
        if (i == row) acc += contrib[(size_t)r * width + w];
    }
    out[(size_t)row * stride + w] += acc;
}

// ---- weight gradients that follow from the bottleneck factors (fp32, see BottWgradArgs).
//      blocks 0..255: bottleneck layer, block = input feature i of the heads' first layers = output feature (row) of the bottleneck
//      layer, thread = column j;  blocks 256..: one row m of [W_A1; W_T1] each, thread = bottleneck feature i ----
constexpr int BOTT_LDS_F = 256 + 256 * 17;
// body of one (virtual) block of 256 threads: vblk = block index, j = thread; s_w [256], s_t [256][17].  Every barrier is executed by
// all threads of the REAL block (two virtual blocks of the same branch share one in the merged tail kernel).
#ifndef EO_TAIL_SKIP      // diagnostic builds only (results WRONG): bit 0 / 1 / 2 drop the bottleneck-row / head-row / embedding roles of the tail kernel
#define EO_TAIL_SKIP 0
#endif
EO_DEV void bott_wgrad_body(const BottWgradArgs& a, int vblk, int j, float* s_w, float (*s_t)[17]) {
    if ((EO_TAIL_SKIP & 1) && vblk < 256) return;
    if ((EO_TAIL_SKIP & 2) && vblk >= 256) return;
    if (vblk < 256) {
        const int i = vblk;
        float acc = 0.f, accb = 0.f;
        if (j < 128) s_w[j] = a.w_a1[j * 256 + i];
        else s_w[j] = a.w_t1 ? a.w_t1[(j - 128) * 260 + i] : 0.f;
        __syncthreads();
        // (unrolled: the 128 loads of a column are independent, 64 of them in flight instead of the few the compiler keeps by itself;
        //  four partial sums keep the adds off one dependency chain)
        float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 64
        for (int k = 0; k < 128; ++k) p4[k & 3] += s_w[k] * a.m_a[k * 256 + j];
        if (a.w_t1) {
#pragma unroll 64
            for (int k = 0; k < 128; ++k) p4[k & 3] += s_w[128 + k] * a.m_t[k * 256 + j];
        }
        acc = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        a.d_w[i * 256 + j] += acc;                      // the only writer of this block of the gradient buffer
        // bias: sum_k W_AT[k][i] db_AT[k] -- one term per thread and a fixed-order tree (a single thread walking the 256 terms cost
        // this role 2 us)
        accb = (j < 128 || a.w_t1) ? s_w[j] * a.db_at[j] : 0.f;
        accb = wave_sum(accb);
        if ((j & 63) == 0) s_t[0][j >> 6] = accb;
        __syncthreads();
        if (j == 0) a.d_b[i] += (s_t[0][0] + s_t[0][1]) + (s_t[0][2] + s_t[0][3]);
        return;
    }
    // first layer of a head, row m:  dW[m][i] += sum_j M[m][j] W_bott[i][j] + db[m] b_bott[i];  thread = i.  W_bott is read through its
    // TRANSPOSED copy (w_bott_t[j][i]: coalesced over i, no staging, no barriers; until round 5 this role walked 16 LDS-transposed tiles of
    // W_bott behind 32 barriers and was the longest chain of the tail kernel, 21 us of its 25).  One accumulator, j ascending: the sums of
    // the tiled version, bit for bit; the loads run 64 ahead of the adds.
    const int mm = vblk - 256, m = mm & 127;
    const bool tr = mm >= 128;
    const float* M = (tr ? a.m_t : a.m_a) + m * 256;
    s_w[j] = M[j];
    const float dbm = a.db_at[mm];
    const float bb = a.b_bott[j];
    float* dw = (tr ? a.d_w_t1 + m * 260 : a.d_w_a1 + m * 256) + j;
    const float old = *dw;
    __syncthreads();
    float acc = 0.f;
#pragma unroll 1
    for (int j0 = 0; j0 < 256; j0 += 64) {
        float w[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) w[c] = a.w_bott_t[(j0 + c) * 256 + j];
#pragma unroll
        for (int c = 0; c < 64; ++c) acc += s_w[j0 + c] * w[c];
    }
    acc += dbm * bb;
    *dw = old + acc;                                    // the only writer of this row in this launch
    if (j == 0) (tr ? a.d_b_t1 : a.d_b_a1)[m] += dbm;
}
__global__ __launch_bounds__(256) void k_bott_wgrad(BottWgradArgs a) {
    __shared__ float s_w[256];
    __shared__ float s_t[256][17];
    bott_wgrad_body(a, blockIdx.x, threadIdx.x, s_w, s_t);
}

// ---- transient embedding gradient: per-sample d emb (from the backward chain) summed per ray, added per image ---
// body of one (virtual) block of 256 threads; s_de: [n_img][4] block-local accumulation (unused when lds_images == 0)
constexpr int EMB_RPW = 4, EMB_RAYS_PER_VBLOCK = 4 * EMB_RPW;      // rays per wave / per 256-thread (virtual) block of the embedding gradient
template <int SPL>
EO_DEV void emb_grad_body_spl(const EmbGradArgs& a, int vblk, int tid, float* s_de) {
    const int lane = tid & 63, ray = vblk * EMB_RAYS_PER_VBLOCK + (tid >> 6);
    const bool lds_acc = a.lds_images > 0 && !a.d_emb_rays;
    if (lds_acc) {
        for (int i = tid; i < a.lds_images * 4; i += 256) s_de[i] = 0.f;
        __syncthreads();
    }
    // 16 rays per block, 4 per wave (8 until round 5: twice the blocks start on twice the CUs, and the role is a chain of HBM round trips).
    // Two passes without early exits, so that the (offset, count, image) loads and then the gradient loads of a wave are in flight together
    // (one dependent chain per ray otherwise)
    int off8[EMB_RPW], n8[EMB_RPW], img8[EMB_RPW];
#pragma unroll
    for (int k8 = 0; k8 < EMB_RPW; ++k8) {
        const int rr = ray + 4 * k8, rc = rr < a.n_rays ? rr : a.n_rays - 1;
        off8[k8] = a.offsets[rc];
        n8[k8] = rr < a.n_rays ? a.counts[rc] : 0;
        img8[k8] = a.d_emb_rays ? 0 : (int)a.img_idx[rc];      // (with the first batch: loaded ray by ray it was eight more round trips)
    }
    f32x4 v8[EMB_RPW][SPL];
#pragma unroll
    for (int k8 = 0; k8 < EMB_RPW; ++k8)
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int i = lane + 64 * k;
            v8[k8][k] = i < n8[k8] ? *reinterpret_cast<const f32x4*>(a.g_emb + 4 * (size_t)(off8[k8] + i)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int k8 = 0; k8 < EMB_RPW; ++k8) {
        const int rr = ray + 4 * k8;
        if (rr >= a.n_rays) break;
        float acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v8[k8][0][e];
#pragma unroll
            for (int k = 1; k < SPL; ++k) t += v8[k8][k][e];
            acc[e] = wave_sum(t);
        }
        if (lane < 4) {
            const float v = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
            if (a.d_emb_rays) a.d_emb_rays[(size_t)rr * 4 + lane] = v;
            else if (lds_acc) atomicAdd(&s_de[img8[k8] * 4 + lane], v);
            else atomicAdd(a.d_emb + img8[k8] * 4 + lane, v);
        }
    }
    if (lds_acc) {
        __syncthreads();
        for (int i = tid; i < a.lds_images * 4; i += 256) { const float v = s_de[i]; if (v != 0.f) atomicAdd(a.d_emb + i, v); }
    }
}
EO_DEV void emb_grad_body(const EmbGradArgs& a, int vblk, int tid, float* s_de) {
    if (a.n_samples <= 64) emb_grad_body_spl<1>(a, vblk, tid, s_de);
    else if (a.n_samples > 128) emb_grad_body_spl<4>(a, vblk, tid, s_de);
    else emb_grad_body_spl<2>(a, vblk, tid, s_de);
}
__global__ __launch_bounds__(256) void k_emb_grad(EmbGradArgs a) {
    extern __shared__ float s_de[];
    emb_grad_body(a, blockIdx.x, threadIdx.x, s_de);
}

// ---- the tail of a training backward in ONE launch: the three kernels above are independent of each other (bottleneck-factor
//      products | embedding table | ambient head), each a few dozen workgroups of 10-35 us -- run one after the other they cost their
//      sum (~70 us of a 3.4 ms step), in one grid the longest.  Blocks of 512 threads: [0, n_amb) ambient head; then pairs of 256-thread
//      virtual blocks of the bottleneck products (a pair never straddles the kernel's two branches: 256 is even) and of the embedding
//      gradient.  n_bott / n_emb / n_amb = 0: that part is absent. ----
struct StepTailArgs { BottWgradArgs bott; EmbGradArgs emb; AmbientBwdArgs amb; EncPartReduceArgs enc; int n_amb, n_bott, n_emb, n_enc; };
// the shadow pass' encoding products (eonerf_enc_pair.hip): sum of the workgroups' partials -> the gradient buffer.  Element e of
// [source 2][row 256][slot 64] | [256] bias sums is summed by ENC_RED_SPLIT threads (one per residue class of the partial's index), 8 loads in
// flight each, and added atomically (4 adders per address: no contention to speak of; every other writer of these elements -- the GEMM
// launch's camera jobs -- has finished).  One thread per element over all 256 partials was 64 dependent HBM round trips: 23 us, the longest role of the launch
constexpr int ENC_RED_SPLIT = 4;
EO_DEV void enc_part_reduce_body(const EncPartReduceArgs& a, int vblk, int tid) {
    const int q = vblk % ENC_RED_SPLIT, e = (vblk / ENC_RED_SPLIT) * 512 + tid;
    if (e >= ENC_PART_F) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = q;
    for (; w + 7 * ENC_RED_SPLIT < a.n_wg; w += 8 * ENC_RED_SPLIT) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += a.part[(size_t)(w + u * ENC_RED_SPLIT) * ENC_PART_F + e];
    }
    for (; w < a.n_wg; w += ENC_RED_SPLIT) s[0] += a.part[(size_t)w * ENC_PART_F + e];
    const float sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (e >= 2 * 256 * 64) { atomicAdd(a.db0 + (e - 2 * 256 * 64), sum); return; }
    const int src = e >> 14, row = (e >> 6) & 255, cm = a.col_map[e & 63];
    if (cm < 0) return;
    if (src == 0) atomicAdd(a.dw0 + row * 63 + cm, sum); else atomicAdd(a.dw5s + row * 319 + cm, sum);
}
__global__ __launch_bounds__(512) void k_step_tail(StepTailArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[AMB_LDS_F];
    static_assert(2 * BOTT_LDS_F <= AMB_LDS_F, "two virtual blocks of the bottleneck products share the ambient staging area");
    int blk = blockIdx.x;
    if (blk < a.n_amb) { ambient_bwd_body(a.amb, blk, a.n_amb, lds); return; }
    blk -= a.n_amb;
    if (blk < a.n_enc) { enc_part_reduce_body(a.enc, blk, threadIdx.x); return; }
    blk -= a.n_enc;
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
    if (blk < a.n_bott / 2) {
        float* base = lds + half * BOTT_LDS_F;
        bott_wgrad_body(a.bott, 2 * blk + half, tid, base, reinterpret_cast<float (*)[17]>(base + 256));
        return;
    }
    blk -= a.n_bott / 2;
    // (an odd last virtual block runs with an idle partner: rays beyond n_rays are skipped inside)
    // (measured, round 5: these blocks behind the bottleneck products in the SAME workgroups -- one round of 256 instead of 384 workgroups on
    //  256 CUs -- is slower, 23.9 vs 21.4 us: the embedding role is three dependent HBM round trips, better started on a free CU)
    if (EO_TAIL_SKIP & 4) return;
    emb_grad_body(a.emb, 2 * blk + half, tid, lds + half * (AMB_LDS_F / 2));
}

// ---- training loss and its gradient on the packed outputs (train_eonerf.py:139-143) ------------------------------
//   kind 0: F.mse_loss(rgb, gt)                          kind 1: metrics.uncertainty_aware_loss (metrics.py:17-22)
//   The mean is summed in a FIXED order without a zeroed accumulator: every block leaves its partial sum in scratch[block], takes a
//   ticket, and the last block to arrive adds the partials in block order, writes the scalar and resets the ticket counter
//   (scratch[LOSS_MAX_BLOCKS], as int) for the next call.
__global__ __launch_bounds__(256) void k_loss(const float* out, const float* gt, int n, int kind, float* d_out, float* loss, float* scratch) {
    __shared__ float s_part[4];
    __shared__ int s_last;
    float part = 0.f;
    for (int ray = blockIdx.x * 256 + threadIdx.x; ray < n; ray += gridDim.x * 256) {
        const float* o = out + (size_t)ray * 21;
        float* d = d_out + (size_t)ray * 21;
#pragma unroll
        for (int c = 0; c < 21; ++c) d[c] = 0.f;
        const float inv = 1.f / (3.f * n);
        if (kind == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float df = o[c] - gt[(size_t)ray * 3 + c]; part += df * df * inv; d[c] = 2.f * df * inv; }
        } else {
            const float beta = o[12], ib2 = 1.f / (beta * beta);
            float sq = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float df = o[c] - gt[(size_t)ray * 3 + c]; sq += df * df; d[c] = df * ib2 * inv; }
            part += 0.5f * sq * ib2 * inv + 0.5f * logf(beta) / n;
            d[12] = -sq * ib2 / beta * inv + 0.5f / (n * beta);
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        __threadfence();
        int* ticket = reinterpret_cast<int*>(scratch + LOSS_MAX_BLOCKS);
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    float total = kind == 1 ? 1.5f : 0.f;                 // the constant 3/2 of the beta term (metrics.py:20)
    for (unsigned b = 0; b < gridDim.x; ++b) total += __hip_atomic_load(scratch + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *loss = total;
    *reinterpret_cast<int*>(scratch + LOSS_MAX_BLOCKS) = 0;
}
// ---- torch.optim.Adam (no weight decay, no amsgrad) on the flat buffers --------------------------------------
// ONE step count for every parameter: while epoch_idx < 2 the reference's graph still reaches the transient / ambient heads through
// torch.cat + slicing (sat_rendering.py:294,311-312,322), so they receive defined ZERO gradients, torch.optim.Adam creates their
// state at step 1 and moves them by 0 -- which is what zero entries of g do here.
// Fault gate: the update is skipped as a whole (parameters and both moments untouched) when the context's sticky status word is set
// (a watchdog of the pipelined backward fired on this rank: its gradients are invalid) or when the reduced fault flag of the
// gradient message is non-zero (some rank's are); the latter also raises the local status word so that this rank's next
// eonerf_device_status reports it.
// ZERO: the gradient is consumed -- optimizer.step() and the optimizer.zero_grad() of the next iteration (train_eonerf.py:158-161) in
// one pass; also on a skipped step (the invalid gradients must not leak into the next accumulation).
template <bool ZERO>
__global__ void k_adam(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, float gscale, int* status, const float* fault_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool remote = fault_flag != nullptr && *fault_flag != 0.f;
    if (remote && i == 0) atomicOr(status, 0x100);
    const bool skip = remote || (status != nullptr && *status != 0);
    if (i >= n) return;
    const float graw = skip ? 0.f : g[i];
    if (ZERO) g[i] = 0.f;
    if (skip) return;
    const float gi = graw * gscale;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);            // exp_avg.lerp_(grad, 1-beta1)
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
// the fault flag of the gradient message (eonerf_grad_seal): control floats [0..3] behind the gradients
__global__ void k_grad_seal(float* tail, const int* status) {
    if (threadIdx.x < 4) tail[threadIdx.x] = (threadIdx.x == 0 && *status != 0) ? 1.f : 0.f;
}

}  // namespace

hipError_t eo_launch_shade_bwd(const ShadeBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_shade_bwd, dim3((a.n_rays + 255) / 256), dim3(256), (size_t)a.lds_images * 6 * sizeof(float), st, a);
    return hipGetLastError();
}
hipError_t eo_launch_rendering_out_bwd(const RenderingOutBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_rendering_out_bwd, dim3((a.n_rays + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_sun_composite_bwd(const CompositeBwdArgs& a, hipStream_t st) {
    eo_dispatch_spl(a.n_samples, [&](auto spl) {
        hipLaunchKernelGGL(k_sun_composite_bwd<decltype(spl)::value>, dim3((a.n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), dim3(256), 0, st, a);
    });
    return hipGetLastError();
}
hipError_t eo_launch_cam_composite_bwd(const CompositeBwdArgs& a, hipStream_t st) {
    eo_dispatch_spl(a.n_samples, [&](auto spl) {
        hipLaunchKernelGGL(k_cam_composite_bwd<decltype(spl)::value>, dim3((a.n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK), dim3(256), 0, st, a);
    });
    return hipGetLastError();
}
hipError_t eo_launch_field_grads_to_soa(const float* g_sigma, const float* g_albedo, const float* g_ts, const float* g_tb, int n, int p_pad,
                                       float* o_sigma, float* o_albedo, float* o_ts, float* o_tb, hipStream_t st) {
    hipLaunchKernelGGL(k_field_grads_to_soa, dim3((n + 255) / 256), dim3(256), 0, st, g_sigma, g_albedo, g_ts, g_tb, n, p_pad, o_sigma, o_albedo, o_ts, o_tb);
    return hipGetLastError();
}
hipError_t eo_launch_emb_grad_points(const float* g_emb, const int* simg, int n, float* d_emb, hipStream_t st) {
    hipLaunchKernelGGL(k_emb_grad_points, dim3((n + 255) / 256), dim3(256), 0, st, g_emb, simg, n, d_emb);
    return hipGetLastError();
}
hipError_t eo_launch_ambient_points_bwd(const AmbientW& w, const float* sun, const float* g_amb, int n,
                                        float* d_w1, float* d_b1, float* d_w2, float* d_b2, hipStream_t st) {
    const int blocks = n < 256 ? (n + 3) / 4 : 64;
    hipLaunchKernelGGL(k_ambient_points_bwd, dim3(blocks), dim3(256), 0, st, w, sun, g_amb, n, d_w1, d_b1, d_w2, d_b2);
    return hipGetLastError();
}
hipError_t eo_launch_bott_wgrad(const BottWgradArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_bott_wgrad, dim3(256 + 128 + (a.w_t1 ? 128 : 0)), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_ambient_bwd(const AmbientBwdArgs& a, hipStream_t st, bool deterministic) {
    static int blocks = 0;
    if (!blocks) { const char* e = getenv("EONERF_AMB_BLOCKS"); blocks = e && atoi(e) > 0 ? atoi(e) : 32; }      // measured (4096 rays): 64 blocks 46 us, 32: 37, 16: 39, 8: 58
    // deterministic mode: ONE block -- every address is then added by exactly one thread, its rays in a fixed order
    const int want = deterministic ? 1 : (a.n_rays + AMB_BATCH - 1) / AMB_BATCH;
    hipLaunchKernelGGL(k_ambient_bwd, dim3(want < blocks ? want : blocks), dim3(128 * AMB_STREAMS), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_step_tail(const BottWgradArgs* bott, const EmbGradArgs* emb, const AmbientBwdArgs* amb, hipStream_t st, const EncPartReduceArgs* enc) {
    StepTailArgs a;
    memset(&a, 0, sizeof(a));
    if (bott) { a.bott = *bott; a.n_bott = 256 + 128 + (bott->w_t1 ? 128 : 0); }
    if (emb) {
        a.emb = *emb;
        a.n_emb = (emb->n_rays + EMB_RAYS_PER_VBLOCK - 1) / EMB_RAYS_PER_VBLOCK;
        if (a.emb.lds_images * 4 > AMB_LDS_F / 2) a.emb.lds_images = 0;      // table too large for the shared staging area: direct atomics
    }
    if (amb) {
        a.amb = *amb;
        const int want = (amb->n_rays + AMB_BATCH - 1) / AMB_BATCH;
        a.n_amb = want < 32 ? want : 32;                                      // (block count: see eo_launch_ambient_bwd)
    }
    if (enc) { a.enc = *enc; a.n_enc = (ENC_PART_F + 511) / 512 * ENC_RED_SPLIT; }
    const int grid = a.n_amb + a.n_enc + a.n_bott / 2 + (a.n_emb + 1) / 2;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(k_step_tail, dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}
hipError_t eo_launch_table_reduce(const float* contrib, const int64_t* idx, int n_rays, int width, int stride, int n_rows, int eval_first,
                                  float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_table_reduce, dim3((n_rows * width + 63) / 64), dim3(64), 0, st, contrib, idx, n_rays, width, stride, n_rows, eval_first, out);
    return hipGetLastError();
}
hipError_t eo_launch_emb_grad(const EmbGradArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_emb_grad, dim3((a.n_rays + EMB_RAYS_PER_VBLOCK - 1) / EMB_RAYS_PER_VBLOCK), dim3(256), (size_t)a.lds_images * 4 * sizeof(float), st, a);
    return hipGetLastError();
}
hipError_t eo_launch_loss(const float* out, const float* gt, int n, int kind, float* d_out, float* loss, float* scratch, hipStream_t st) {
    const int blocks = (n + 255) / 256 < LOSS_MAX_BLOCKS ? (n + 255) / 256 : LOSS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_loss, dim3(blocks), dim3(256), 0, st, out, gt, n, kind, d_out, loss, scratch);
    return hipGetLastError();
}
hipError_t eo_launch_adam(float* p, float* g, bool zero_grad, float* m, float* v, size_t n, int step, float lr, float b1, float b2, float eps,
                          float gscale, int* status, const float* fault_flag, hipStream_t st) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (zero_grad) hipLaunchKernelGGL(k_adam<true>, grid, dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2), gscale, status, fault_flag);
    else hipLaunchKernelGGL(k_adam<false>, grid, dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2), gscale, status, fault_flag);
    return hipGetLastError();
}
hipError_t eo_launch_grad_seal(float* tail, const int* status, hipStream_t st) {
    hipLaunchKernelGGL(k_grad_seal, dim3(1), dim3(64), 0, st, tail, status);
    return hipGetLastError();
}
