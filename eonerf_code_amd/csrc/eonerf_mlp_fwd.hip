// Fused EO-NeRF field forward (H4+H5+H6 of SURVEY.md 8a):
//   positional encoding -> 8x256 trunk (skip-concat after layer 4) -> sigma | albedo head, transient head
//   (radiance_fields/eonerf.py:154-170, radiance_fields/mlp.py:87-101,190-208); the identity-activation bottleneck layer between the
//   trunk and the heads is folded into the heads' first layers (eonerf_pack.h).
// One persistent workgroup walks tiles of P::TILE samples.  Activations never leave registers (transposed
// H^T chain, see eonerf_common.h); weights stream L2 -> LDS in packed fragment order (eonerf_pack.cpp).
// TRAIN additionally saves, per layer, the post-activation tensor feature-major [F][P_pad] (the B operand
// of the weight-gradient GEMM) and 1 bit/element ReLU masks (for the backward chain).
#include "eonerf_common.h"
#include "eonerf_kernels.h"

#ifdef EO_STAMP
__device__ unsigned long long eo_stamps_fwd[8];
#endif
#ifndef EO_MASK_NT
#define EO_MASK_NT 1
#endif
namespace {

template <class P> struct EncUnits { typename P::U u[ENC_SLOTS / P::KF]; };

template <class P>
EO_DEV EncUnits<P> encode_position(float x, float y, float z, int h) {
    EncUnits<P> E;
    const float off = h ? EO_PI_2_F : 0.0f;
    constexpr int NU = ENC_SLOTS / P::KF;
#pragma unroll
    for (int kg = 0; kg < NU; ++kg)
#pragma unroll
        for (int e = 0; e < P::NE; ++e) {
            const int q = kg * P::NE + e;
            float v;
            if (q < 30) {
                const int k = q / 3, d = q % 3;
                const float c = d == 0 ? x : (d == 1 ? y : z);
                const float arg = c * (float)(1 << k) + off;     // exact scale, one fp32 add (mlp.py:199-203)
                v = P::IS_BF16 ? __sinf(arg) : sinf(arg);
            } else if (q == 30) {
                v = h ? z : x;
            } else {
                v = h ? 0.0f : y;
            }
            set_elem(E.u[kg], e, v);
        }
    return E;
}

// MODE 0: inference.  1: training, everything saved.  2: training with the transient head OUTSIDE the autograd graph
// (epoch_idx < 2: s = 1 and the loss is MSE on rgb -- train_eonerf.py:139-141, sat_rendering.py:269-272): its forward still
// runs (ts / beta are outputs) but nothing of it is saved.
// TMASK = false: the ReLU bits of trunk layers 0..6 are not needed (the pipelined trunk backward derives ReLU' from the saved
// activations themselves, a.mask_from >= 7): their two VALU operations per epilogue slice are compiled out.
template <class P, bool FULL, int MODE, bool TMASK = true>
__global__ __launch_bounds__(P::NT) void k_mlp_fwd(MlpFwdArgs a) {
    constexpr bool TRAIN = MODE != 0, TSAVE = MODE == 1;
    constexpr int SLOT = FwdSlot<P>::BYTES;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef typename P::U U;
    constexpr int HKG = 256 / P::KF;      // k-groups of a 256-wide activation
    constexpr int QKG = 128 / P::KF;      // ... of a 128-wide activation
    constexpr int EKG = ENC_SLOTS / P::KF;
    constexpr int NST = TRAIN ? (P::IS_BF16 ? 2 : 16) : 0;     // slab stores per m-tile epilogue (SlabWriter::tile)
    constexpr int NST_T = TSAVE ? NST : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int n_pts = *a.n_pts;

    WStream<P, SLOT> ws;
    ws.g = a.stream; ws.tab = a.chunks; ws.lds = smem; ws.n_chunks = a.n_chunks;
    ws.wave_b = __builtin_amdgcn_readfirstlane(wave) * 1024; ws.lane_b = lane * 16;
    ws.late = P::WAVES == 8 && (a.stagger == 1 ? __builtin_amdgcn_readfirstlane(wave) >= 4 : (a.stagger == 2 && (__builtin_amdgcn_readfirstlane(wave) & 1)));
    const TileSched<P> sched(n_pts, gridDim.x, blockIdx.x);
    if (sched.iters() == 0) return;      // uniform per workgroup
    ws.start();
#ifdef EO_STAMP
    const unsigned long long t_begin = EO_T();
#endif

    for (int it = 0; it < sched.iters(); ++it) {
        const int wt = sched.first(it) + __builtin_amdgcn_readfirstlane(wave);      // this wave's 32 samples
        if (__builtin_amdgcn_readfirstlane(wave) >= sched.waves(it)) { ws.idle_tile(); continue; }
        const int p = wt * 32 + c;          // this lane's sample (both halves share it)
        const bool live = p < n_pts;
        const float x = live ? a.px[p] : 0.f, y = live ? a.py[p] : 0.f, z = live ? a.pz[p] : 0.f;
        const EncUnits<P> E = encode_position<P>(x, y, z, h);
        SlabWriter<P, ActMap> sw;                                                   // this wave's sample tile(s) of the activation slab
        if constexpr (TRAIN) {
            sw.init(a.act, a.p_pad / Slab<P>::TSAMP, wt * 32, lane, smem + 2 * SLOT + wave * 2 * TR_WAVE_B);
#pragma unroll
            for (int kg = 0; kg < EKG; ++kg)      // encoding slots, rows [0,64)
#pragma unroll
                for (int e = 0; e < P::NE; ++e) sw.elem(ACT_ROW_ENC + P::feat(kg, 0, e), get_elem(E.u[kg], e));
        }

        auto& mid = sw;          // run_layer's slab-flush hooks
        U H[HKG], N[HKG];
        uint32_t mbits[4];
        if constexpr (P::IS_SPLIT) {
            // split policy: mbits[0] = wave-uniform "an operand of this sample tile left fp16's range" flag (mask_commit(PH3)); the raw
            // coordinates are operands of layer 0 and of the skip layer
            mbits[0] = __builtin_amdgcn_ballot_w64(!(fabsf(x) <= 65504.f && fabsf(y) <= 65504.f && fabsf(z) <= 65504.f)) != 0ull ? 1u : 0u;
        }
        uint32_t tbits = 0;      // ReLU flags of the tile whose epilogue is in progress

        // ---------------- trunk ----------------
        // slice s of the epilogue of m-tile mt (see chunk_compute): ReLU, mask flags, next layer's operand, slab staging
        auto relu_epi = [&](U* dst, int act_row, int mt, const f32x16& accv, int s) {
            const Sl<P> v = relu_slice(P(), accv, s, tbits);
            put_slice(P(), dst, mt, s, v);
            if (s == EPI_SLICES - 1) mask_commit(P(), mt, tbits, mbits[P::IS_SPLIT ? 0 : mt >> 1]);
            if constexpr (TRAIN) sw.stage(act_row + 32 * mt, s, v);
        };
        auto relu_epi_nomask = [&](U* dst, int act_row, int mt, const f32x16& accv, int s) {      // trunk layers 0..6 with TMASK = false
            const Sl<P> v = relu_only_slice(P(), accv, s);
            put_slice(P(), dst, mt, s, v);
            if constexpr (TRAIN) sw.stage(act_row + 32 * mt, s, v);
        };
        auto relu_epi_t = [&](U* dst, int act_row, int mt, const f32x16& accv, int s) {      // transient head layers
            const Sl<P> v = relu_slice(P(), accv, s, tbits);
            put_slice(P(), dst, mt, s, v);
            if (s == EPI_SLICES - 1) mask_commit(P(), mt, tbits, mbits[P::IS_SPLIT ? 0 : mt >> 1]);
            if constexpr (TSAVE) sw.stage(act_row + 32 * mt, s, v);
        };
        auto save_mask = [&](int mask_slot, int nwords) {
            if constexpr (TRAIN) {
                if (mask_slot < a.mask_from) return;      // wave-uniform
                uint32_t* mp = a.masks + ((size_t)mask_slot * a.p_pad * 2 + (size_t)p * 2 + h) * 4;
#if EO_MASK_NT      // written once, read once by the backward chain long after the caches have turned over: streaming, like the slab stores
                if (nwords == 4) __builtin_nontemporal_store(u32x4{mbits[0], mbits[1], mbits[2], mbits[3]}, reinterpret_cast<u32x4*>(mp));
                else __builtin_nontemporal_store(u32x2{mbits[0], mbits[1]}, reinterpret_cast<u32x2*>(mp));
#else
                if (nwords == 4) *reinterpret_cast<u32x4*>(mp) = u32x4{mbits[0], mbits[1], mbits[2], mbits[3]};
                else *reinterpret_cast<u32x2*>(mp) = u32x2{mbits[0], mbits[1]};
#endif
            }
        };
        auto plain_layer = [&](U* src, U* dst, int l) {
            if (TMASK || l == 7) {
                run_layer<P, SLOT, HKG, 8, true, NST>(ws, mid, lane, h, [&](int kg) { return src[kg]; },
                    [&](int mt, const f32x16& v, int s) { relu_epi(dst, ACT_ROW_X1 + 256 * l, mt, v, s); });
                save_mask(l, 4);
            } else {
                run_layer<P, SLOT, HKG, 8, true, NST>(ws, mid, lane, h, [&](int kg) { return src[kg]; },
                    [&](int mt, const f32x16& v, int s) { relu_epi_nomask(dst, ACT_ROW_X1 + 256 * l, mt, v, s); });
            }
        };

        // layer 0: enc(64) -> 256
        if constexpr (TMASK) {
            run_layer<P, SLOT, EKG, 8, true, NST>(ws, mid, lane, h, [&](int kg) { return E.u[kg]; },
                [&](int mt, const f32x16& v, int s) { relu_epi(H, ACT_ROW_X1, mt, v, s); });
            save_mask(0, 4);
        } else {
            run_layer<P, SLOT, EKG, 8, true, NST>(ws, mid, lane, h, [&](int kg) { return E.u[kg]; },
                [&](int mt, const f32x16& v, int s) { relu_epi_nomask(H, ACT_ROW_X1, mt, v, s); });
        }
        plain_layer(H, N, 1);
        plain_layer(N, H, 2);
        plain_layer(H, N, 3);
        plain_layer(N, H, 4);
        // layer 5 consumes [h, enc] (skip-concat after layer 4, mlp.py:92-97)
        if constexpr (TMASK) {
            run_layer<P, SLOT, HKG + EKG, 8, true, NST>(ws, mid, lane, h,
                [&](int kg) { return kg < HKG ? H[kg < HKG ? kg : 0] : E.u[kg >= HKG ? kg - HKG : 0]; },
                [&](int mt, const f32x16& v, int s) { relu_epi(N, ACT_ROW_X1 + 256 * 5, mt, v, s); });
            save_mask(5, 4);
        } else {
            run_layer<P, SLOT, HKG + EKG, 8, true, NST>(ws, mid, lane, h,
                [&](int kg) { return kg < HKG ? H[kg < HKG ? kg : 0] : E.u[kg >= HKG ? kg - HKG : 0]; },
                [&](int mt, const f32x16& v, int s) { relu_epi_nomask(N, ACT_ROW_X1 + 256 * 5, mt, v, s); });
        }
        plain_layer(N, H, 6);
        plain_layer(H, N, 7);
        // after l=7 (odd) the trunk output X8 lives in N
        // ---------------- sigma (+ heads) ----------------
        float sigma_raw = 0.f;
        if constexpr (!FULL) {
            run_layer<P, SLOT, HKG, 1, true>(ws, mid, lane, h, [&](int kg) { return N[kg]; },
                [&](int, const f32x16& v, int s) { if (s == 0) sigma_raw = v[0]; });
            if (h == 0 && live) a.sigma[p] = softplus_f(sigma_raw);
        } else {
            // The bottleneck layer (identity activation) is never evaluated: both heads' first layers read X_8 through weights FOLDED
            // with it (eonerf_pack.h).  m-tiles 0..3 = albedo hidden: 256 -> 128 (ReLU), m-tile 4 = sigma row
            U A1[QKG];
            run_layer<P, SLOT, HKG, 5, true, NST>(ws, mid, lane, h, [&](int kg) { return N[kg]; },
                [&](int mt, const f32x16& v, int s) {
                    if (mt == 4) { if (s == 0) sigma_raw = v[0]; return; }
                    relu_epi(A1, ACT_ROW_A1, mt, v, s);
                });
            save_mask(8, 2);
            if (h == 0 && live) a.sigma[p] = softplus_f(sigma_raw);

            // ---------------- albedo head, output layer: 128 -> 3 (Sigmoid) ----------------
            run_layer<P, SLOT, QKG, 1, true>(ws, mid, lane, h, [&](int kg) { return A1[kg]; },
                [&](int, const f32x16& v, int s) {
                    if (s == 0 && h == 0 && live) {
                        a.albedo[p] = sigmoid_f(v[0]);
                        a.albedo[(size_t)a.p_pad + p] = sigmoid_f(v[1]);
                        a.albedo[2 * (size_t)a.p_pad + p] = sigmoid_f(v[2]);
                    }
                });

            // ---------------- transient head: [bottleneck(X_8), emb(img)] (260) -> 4x128 (ReLU) -> {Sigmoid, Softplus} ----
            U EMB = P::zero();
            if (h == 0) {
                const int im = live ? a.simg[p] : 0;
                const f32x4 ev = *reinterpret_cast<const f32x4*>(a.emb + 4 * im);
#pragma unroll
                for (int e = 0; e < 4; ++e) set_elem(EMB, e, ev[e]);
            }
            if constexpr (TSAVE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sw.elem(ACT_ROW_EMB + e, get_elem(EMB, e));
            }
            U T1[QKG], T2[QKG];
            run_layer<P, SLOT, HKG + 1, 4, true, NST_T>(ws, mid, lane, h,
                [&](int kg) { return kg < HKG ? N[kg < HKG ? kg : 0] : EMB; },
                [&](int mt, const f32x16& v, int s) { relu_epi_t(T1, ACT_ROW_T1, mt, v, s); });
            if constexpr (TSAVE) save_mask(9, 2);
            run_layer<P, SLOT, QKG, 4, true, NST_T>(ws, mid, lane, h, [&](int kg) { return T1[kg]; },
                [&](int mt, const f32x16& v, int s) { relu_epi_t(T2, ACT_ROW_T1 + 128, mt, v, s); });
            if constexpr (TSAVE) save_mask(10, 2);
            run_layer<P, SLOT, QKG, 4, true, NST_T>(ws, mid, lane, h, [&](int kg) { return T2[kg]; },
                [&](int mt, const f32x16& v, int s) { relu_epi_t(T1, ACT_ROW_T1 + 256, mt, v, s); });
            if constexpr (TSAVE) save_mask(11, 2);
            run_layer<P, SLOT, QKG, 4, true, NST_T>(ws, mid, lane, h, [&](int kg) { return T1[kg]; },
                [&](int mt, const f32x16& v, int s) { relu_epi_t(T2, ACT_ROW_T1 + 384, mt, v, s); });
            if constexpr (TSAVE) save_mask(12, 2);
            run_layer<P, SLOT, QKG, 1, true>(ws, mid, lane, h, [&](int kg) { return T2[kg]; },
                [&](int, const f32x16& v, int s) {
                    if (s == 0 && h == 0 && live) { a.ts[p] = sigmoid_f(v[0]); a.tb[p] = softplus_f(v[1]); }
                });
        }
        if constexpr (TRAIN) sw.drain();
        if constexpr (P::IS_SPLIT) {
            // an activation beyond fp16's largest finite value (65504), or a raw coordinate that is
            if (mbits[0] != 0u && lane == 0) atomicOr(a.range_flag, 1);      // (wave-uniform condition)
        }
    }
#ifdef EO_STAMP
    if (lane == 0) {
        atomicAdd(&eo_stamps_fwd[0], EO_T() - t_begin); atomicAdd(&eo_stamps_fwd[1], ws.t_wait); atomicAdd(&eo_stamps_fwd[2], ws.t_bar); atomicAdd(&eo_stamps_fwd[3], ws.t_flush); atomicAdd(&eo_stamps_fwd[4], 1ull);
    }
#endif
}

template <class P, bool FULL, int MODE, bool TMASK = true>
hipError_t launch(const MlpFwdArgs& a, int grid, hipStream_t st) {
    constexpr int SMEM = 2 * FwdSlot<P>::BYTES + (MODE ? SlabWriter<P, ActMap>::LDS_BYTES : 0);
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fwd<P, FULL, MODE, TMASK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_mlp_fwd<P, FULL, MODE, TMASK>), dim3(grid), dim3(P::NT), SMEM, st, a);
    return hipGetLastError();
}

template <class P> hipError_t dispatch(const MlpFwdArgs& a, bool full, int mode, int grid, hipStream_t st) {
    if (full) return mode == 0 ? launch<P, true, 0>(a, grid, st) : (mode == 1 ? launch<P, true, 1>(a, grid, st) : launch<P, true, 2>(a, grid, st));
    return mode == 0 ? launch<P, false, 0>(a, grid, st) : launch<P, false, 1>(a, grid, st);
}

}  // namespace

// mode: 0 inference, 1 training, 2 training with the transient head outside the autograd graph (full variant only)
// prec: 0 fp32, 1 bf16, 2 fp16 x 3 split (inference only)
hipError_t eo_launch_mlp_fwd(const MlpFwdArgs& a, int prec, bool full, int mode, int grid, hipStream_t st) {
    if (prec == 2) {
        if (mode != 0) return hipErrorInvalidValue;
        return full ? launch<PH3, true, 0>(a, grid, st) : launch<PH3, false, 0>(a, grid, st);
    }
    const bool bf16 = prec == 1;
    if (bf16 && mode != 0 && a.mask_from >= 7) {      // training pass in front of the pipelined trunk backward: no trunk mask bits
        if (!full) return launch<PBf16, false, 1, false>(a, grid, st);
        return mode == 1 ? launch<PBf16, true, 1, false>(a, grid, st) : launch<PBf16, true, 2, false>(a, grid, st);
    }
    return bf16 ? dispatch<PBf16>(a, full, mode, grid, st) : dispatch<PF32>(a, full, mode, grid, st);
}

#ifdef EO_STAMP
extern "C" void eonerf_debug_read_fwd(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(eo_stamps_fwd), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(eo_stamps_fwd), z, sizeof(z));
}
#endif
