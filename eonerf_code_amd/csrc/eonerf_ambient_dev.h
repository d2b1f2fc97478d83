// Ambient-head backward for a range of rays, as a device function shared by the per-ray kernels (eonerf_rays_bwd.hip: k_ambient_bwd,
// k_step_tail) and the pipelined trunk backward (eonerf_bwd_pipe.hip: the workgroups of its launch that have no pipeline stage).
#pragma once
#include "eonerf_common.h"
#include "eonerf_rays.h"

namespace {

// ---- ambient head backward (radiance_fields/eonerf.py:132-139): thread = (hidden unit j, ray stream q).  A block owns a contiguous
//      range of rays and walks it in batches of AMB_BATCH: the batch's records (sun encoding + hidden activations saved by the
//      forward, output and its gradient) are staged in LDS by ONE coalesced pass of the whole block, then every stream takes
//      every AMB_STREAMS-th ray of the batch out of LDS (per-ray global loads made the loop a chain of dependent L2 round trips:
//      0.67 us per ray and thread; from LDS 0.11 us, which is the loop's VALU / LDS instruction time on the few CUs in use).
//      The streams' partial sums meet in LDS before ONE set of atomics per block (every block adds into the same 3.9k addresses:
//      ~0.7 us per block, hence few blocks) ----
constexpr int AMB_STREAMS = 4, AMB_BATCH = 64, AMB_REC = 168;       // per staged ray: 160 saved floats, 3 outputs, 3 gradients, pad
constexpr int AMB_LDS_F = AMB_BATCH * AMB_REC > (AMB_STREAMS - 1) * 128 * 32 ? AMB_BATCH * AMB_REC : (AMB_STREAMS - 1) * 128 * 32;
// body of one ambient block: blk of nblk blocks of 128 * AMB_STREAMS threads; lds: AMB_LDS_F floats, 16-byte aligned
EO_DEV void ambient_bwd_body(const AmbientBwdArgs& a, int blk, int nblk, float* lds) {
#pragma clang fp contract(off)      // the same unfused fp32 arithmetic in every translation unit (eonerf_rays_bwd.hip is built without FMA contraction)
    const int j = threadIdx.x & 127, q = threadIdx.x >> 7;
    float dw1[27], dw2[3] = {0.f, 0.f, 0.f}, db1 = 0.f, db2[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 27; ++i) dw1[i] = 0.f;
    const float w2[3] = {a.w.w2[j], a.w.w2[128 + j], a.w.w2[256 + j]};
    const int per = (a.n_rays + nblk - 1) / nblk;
    const int r_lo = blk * per, r_hi = r_lo + per < a.n_rays ? r_lo + per : a.n_rays;
    // software pipeline: the loads of batch b + 1 (20 + 1 per thread) are in flight while batch b is consumed out of LDS -- one batch
    // at a time would expose a full memory round trip (~2-3 us with so few workgroups on the chip) per 64 rays
    constexpr int NT = 128 * AMB_STREAMS, NL = AMB_BATCH * 160 / NT;
    static_assert(AMB_BATCH * 160 % NT == 0 && AMB_BATCH * 6 <= NT, "staging loop shape");
    float v[NL], v6 = 0.f;
    const int r6 = threadIdx.x / 6, k6 = threadIdx.x - 6 * r6;
    auto fetch = [&](int b0) {
        const int nb = r_hi - b0 < AMB_BATCH ? r_hi - b0 : AMB_BATCH;
        const float* src = a.amb_save + (size_t)b0 * 160;           // the batch's saved records are one contiguous run
#pragma unroll
        for (int u = 0; u < NL; ++u) { const int i = threadIdx.x + u * NT; v[u] = i < nb * 160 ? src[i] : 0.f; }
        v6 = 0.f;
        if (r6 < nb) v6 = k6 < 3 ? a.ray_rec[(size_t)(b0 + r6) * RAY_REC + RR_AMB + k6] : a.g_ray[(size_t)(b0 + r6) * RAY_REC + RR_AMB + k6 - 3];
    };
    if (r_lo < r_hi) fetch(r_lo);
    for (int b0 = r_lo; b0 < r_hi; b0 += AMB_BATCH) {
        const int nb = r_hi - b0 < AMB_BATCH ? r_hi - b0 : AMB_BATCH;
        __syncthreads();                                    // the previous batch is consumed
#pragma unroll
        for (int u = 0; u < NL; ++u) { const int i = threadIdx.x + u * NT, r = i / 160; lds[r * AMB_REC + (i - 160 * r)] = v[u]; }
        if (r6 < AMB_BATCH) lds[r6 * AMB_REC + 160 + k6] = v6;
        __syncthreads();
        if (b0 + AMB_BATCH < r_hi) fetch(b0 + AMB_BATCH);
        for (int r = q; r < nb; r += AMB_STREAMS) {        // rays outside the graph (s == 1: all three gradients zero) add exact zeros
            const float* sv = lds + r * AMB_REC;
            const float hid = sv[32 + j];
            // the ray's shared values as 16-byte broadcast reads
            f32x4 e4[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) e4[i] = *reinterpret_cast<const f32x4*>(sv + 4 * i);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(sv + 160), o1 = *reinterpret_cast<const f32x4*>(sv + 164);
            const float outv[3] = {o0[0], o0[1], o0[2]}, gv[3] = {o0[3], o1[0], o1[1]};
            float gpre[3], ghid = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                gpre[o] = gv[o] * outv[o] * (1.f - outv[o]);
                dw2[o] += gpre[o] * hid;
                db2[o] += gpre[o];
                ghid += w2[o] * gpre[o];
            }
            if (hid <= 0.f) ghid = 0.f;
            db1 += ghid;
#pragma unroll
            for (int i = 0; i < 27; ++i) dw1[i] += ghid * e4[i >> 2][i & 3];
        }
    }
    __syncthreads();                                        // the staging area becomes the reduction area
    // streams 1.. hand their sums to stream 0 through LDS: slots 0..26 dw1, 27 db1, 28..30 dw2
    float (*red)[128][32] = reinterpret_cast<float (*)[128][32]>(lds);
    if (q > 0) {
        float* r = red[q - 1][j];
#pragma unroll
        for (int i = 0; i < 27; ++i) r[i] = dw1[i];
        r[27] = db1; r[28] = dw2[0]; r[29] = dw2[1]; r[30] = dw2[2];
        // db2 is the same for every hidden unit of a stream: units 0..2 hand one component each to stream 0 (slot 31)
        if (j < 3) r[31] = j == 0 ? db2[0] : (j == 1 ? db2[1] : db2[2]);
    }
    __syncthreads();
    if (q > 0) return;
    if (j < 3) {      // one add per component and block, the streams summed in a fixed order
        float v = j == 0 ? db2[0] : (j == 1 ? db2[1] : db2[2]);
#pragma unroll
        for (int s = 0; s < AMB_STREAMS - 1; ++s) v += red[s][j][31];
        if (v != 0.f) atomicAdd(a.d_b2 + j, v);
    }
#pragma unroll
    for (int s = 0; s < AMB_STREAMS - 1; ++s) {
        const float* r = red[s][j];
#pragma unroll
        for (int i = 0; i < 27; ++i) dw1[i] += r[i];
        db1 += r[27]; dw2[0] += r[28]; dw2[1] += r[29]; dw2[2] += r[30];
    }
    // zero contributions are skipped: with the shadow pass off the ambient head is outside the graph (1 - s == 0)
#pragma unroll
    for (int i = 0; i < 27; ++i) if (dw1[i] != 0.f) atomicAdd(a.d_w1 + j * 27 + i, dw1[i]);
    if (db1 != 0.f) atomicAdd(a.d_b1 + j, db1);
#pragma unroll
    for (int o = 0; o < 3; ++o) if (dw2[o] != 0.f) atomicAdd(a.d_w2 + o * 128 + j, dw2[o]);
}

}  // namespace
