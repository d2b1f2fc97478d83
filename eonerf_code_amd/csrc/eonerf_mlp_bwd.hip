// Fused EO-NeRF field backward chain (H10 of SURVEY.md 8a: the autograd of radiance_fields/eonerf.py:154-170).
// Mirror image of eonerf_mlp_fwd.hip: gradients flow through the layers in reverse as TRANSPOSED register tiles
// (dX^T = W^T dY^T: A operand = packed W^T streamed through LDS, B operand = the previous accumulators), ReLU
// derivatives come from the 1-bit masks the forward saved, and every pre-activation gradient dY_l is written
// feature-major [rows][p_pad] -- the A operand of the weight-gradient GEMM (eonerf_wgrad.hip).
// INPUT_GRAD (shadow pass): continues through layer 0 / the skip columns of layer 5 and the encoder derivative
// 2^k cos(2^k x [+ pi/2]) to d sigma / d position (sat_rendering.py:90 keeps depth attached).
#include "eonerf_common.h"
#include "eonerf_kernels.h"

#ifdef EO_STAMP
__device__ unsigned long long eo_stamps_bwd[8];
#endif
#ifndef EO_MASK_NT
#define EO_MASK_NT 1
#endif
#ifndef EO_DY7_NT
#define EO_DY7_NT 1
#endif
namespace {

// k-group whose first (up to 4) features carry `v` on the h==0 lanes (rows 0..3 of a 32-row tile), zero elsewhere
template <class P> EO_DEV typename P::U small_unit(const float (&v)[4], int h) {
    typename P::U u = P::zero();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = h == 0 ? v[e] : 0.f;
        if constexpr (P::IS_BF16) u[e] = (__bf16)x; else u[e] = x;
    }
    return u;
}

// TRANS = false: the transient head is outside the autograd graph (see k_mlp_fwd MODE 2): its layers are skipped.
// PIPE 1 (bf16): the kernel stops at dY_7 and hands it, in B-operand unit order, to the layer-pipelined trunk backward
// (eonerf_bwd_pipe.hip), which computes dY_6..dY_0 AND the trunk weight gradients without parking dY in HBM.
// The bottleneck layer (identity activation) does not appear: the heads' first layers are folded with it (eonerf_pack.h), so
// [dY_A1; dY_T1; d sigma_pre] goes to dX_8 in ONE layer; the bottleneck layer's own weight gradient follows from the bottleneck
// factors (BottWgradArgs).
template <class P, bool FULL, bool IG, bool TRANS, int PIPE = 0>
__global__ __launch_bounds__(P::NT) void k_mlp_bwd(MlpBwdArgs a) {
    constexpr int SLOT = FwdSlot<P>::BYTES;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef typename P::U U;
    constexpr int HKG = 256 / P::KF, QKG = 128 / P::KF;
    constexpr int NST = P::IS_BF16 ? 2 : 16;      // slab stores per m-tile epilogue (SlabWriter::tile)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int n_pts = *a.n_pts;

    WStream<P, SLOT> ws;
    ws.g = a.stream; ws.tab = a.chunks; ws.lds = smem; ws.n_chunks = a.n_chunks;
    ws.wave_b = __builtin_amdgcn_readfirstlane(wave) * 1024; ws.lane_b = lane * 16;
    ws.late = P::WAVES == 8 && (a.stagger == 1 ? __builtin_amdgcn_readfirstlane(wave) >= 4 : (a.stagger == 2 && (__builtin_amdgcn_readfirstlane(wave) & 1)));
    const TileSched<P> sched(n_pts, gridDim.x, blockIdx.x);
    if (sched.iters() == 0) return;
    ws.start();
#ifdef EO_STAMP
    const unsigned long long t_begin = EO_T();
#endif

    for (int it = 0; it < sched.iters(); ++it) {
        const int wt = sched.first(it) + __builtin_amdgcn_readfirstlane(wave);      // this wave's 32 samples (balanced tail: TileSched)
        if (__builtin_amdgcn_readfirstlane(wave) >= sched.waves(it)) { ws.idle_tile(); continue; }
        const int p = wt * 32 + c;
        const bool live = p < n_pts;
        SlabWriter<P, GrdMap> sw;                                                   // this wave's sample tile(s) of the gradient slab
        sw.init(a.grd, a.p_pad / Slab<P>::TSAMP, wt * 32, lane, smem + 2 * SLOT + wave * 2 * TR_WAVE_B);
        uint32_t mb[4];
        auto& mid = sw;          // run_layer's slab-flush hooks

        auto load_mask = [&](int slot, int nwords) {
            const uint32_t* mp = a.masks + ((size_t)slot * a.p_pad * 2 + (size_t)p * 2 + h) * 4;
            if (nwords == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(mp); mb[0] = v[0]; mb[1] = v[1]; mb[2] = v[2]; mb[3] = v[3]; }
            else { const u32x2 v = *reinterpret_cast<const u32x2*>(mp); mb[0] = v[0]; mb[1] = v[1]; }
        };
        // the heads' mask words are fetched at the top of the tile, with the per-sample scalars: several of the head layers have ONE
        // k-group, so a mask loaded in front of its layer is needed a few dozen cycles later -- a whole memory latency exposed per layer
        // in a kernel that is HBM-bound since the bottleneck fold (round 4)
#if EO_MASK_NT
        auto fetch2 = [&](int slot) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(a.masks + ((size_t)slot * a.p_pad * 2 + (size_t)p * 2 + h) * 4)); };
#else
        auto fetch2 = [&](int slot) { return *reinterpret_cast<const u32x2*>(a.masks + ((size_t)slot * a.p_pad * 2 + (size_t)p * 2 + h) * 4); };
#endif
        auto use2 = [&](const u32x2& v) { mb[0] = v[0]; mb[1] = v[1]; };
        // dX tile -> (optional ReLU mask) -> units of the next backward layer + feature-major save for the wgrad GEMM
        // (slice s of the epilogue of m-tile mt, see chunk_compute)
        auto grad_epi = [&](U* dst, int grd_row, bool masked, int mt, const f32x16& accv, int s) {
            const Sl<P> v = masked ? mask_slice(P(), accv, s, mt, mb[mt >> 1]) : pack_slice(P(), accv, s);
            put_slice(P(), dst, mt, s, v);
            sw.stage(grd_row + 32 * mt, s, v);
        };

        // ---------------- output heads: activation derivatives from the saved forward outputs ----------------
        const float sg = live ? a.sigma[p] : 0.f;
        float dsig[4] = {live ? a.g_sigma[p] * (1.f - expf(-sg)) : 0.f, 0.f, 0.f, 0.f};    // softplus' = 1 - exp(-softplus)
        sw.elem(GRD_ROW_SIG, h == 0 ? dsig[0] : 0.f);

        U D[HKG], N[HKG];
        if constexpr (FULL) {
            u32x2 pm12 = {0, 0}, pm11 = {0, 0}, pm10 = {0, 0}, pm9 = {0, 0};
            if constexpr (TRANS) { pm12 = fetch2(12); pm11 = fetch2(11); pm10 = fetch2(10); pm9 = fetch2(9); }
            const u32x2 pm8 = fetch2(8);
            const u32x4 pm7 = *reinterpret_cast<const u32x4*>(a.masks + ((size_t)7 * a.p_pad * 2 + (size_t)p * 2 + h) * 4);
            float dalb[4] = {0.f, 0.f, 0.f, 0.f}, dtr[4] = {0.f, 0.f, 0.f, 0.f};
            if (live) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float av = a.albedo[(size_t)k * a.p_pad + p];
                    dalb[k] = a.g_albedo[(size_t)k * a.p_pad + p] * av * (1.f - av);           // sigmoid'
                }
                if constexpr (TRANS) {
                    const float tsv = a.ts[p], tbv = a.tb[p];
                    dtr[0] = a.g_ts[p] * tsv * (1.f - tsv);
                    dtr[1] = a.g_tb[p] * (1.f - expf(-tbv));
                }
            }
            if constexpr (TRANS) {
#pragma unroll
                for (int e = 0; e < 2; ++e) sw.elem(GRD_ROW_T5 + e, h == 0 ? dtr[e] : 0.f);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) sw.elem(GRD_ROW_A2 + e, h == 0 ? dalb[e] : 0.f);
            U TA[QKG], TB[QKG], DA1[QKG];
            if constexpr (TRANS) {
                // ---- transient head backwards: {ts,tb}_pre -> T4 -> T3 -> T2 -> T1 ----
                const U u_tr = small_unit<P>(dtr, h);
                use2(pm12);
                run_layer<P, SLOT, 1, 4, false, NST>(ws, mid, lane, h, [&](int) { return u_tr; },
                    [&](int mt, const f32x16& v, int s) { grad_epi(TA, GRD_ROW_T1 + 384, true, mt, v, s); });
                use2(pm11);
                run_layer<P, SLOT, QKG, 4, false, NST>(ws, mid, lane, h, [&](int kg) { return TA[kg]; },
                    [&](int mt, const f32x16& v, int s) { grad_epi(TB, GRD_ROW_T1 + 256, true, mt, v, s); });
                use2(pm10);
                run_layer<P, SLOT, QKG, 4, false, NST>(ws, mid, lane, h, [&](int kg) { return TB[kg]; },
                    [&](int mt, const f32x16& v, int s) { grad_epi(TA, GRD_ROW_T1 + 128, true, mt, v, s); });
                use2(pm9);
                run_layer<P, SLOT, QKG, 4, false, NST>(ws, mid, lane, h, [&](int kg) { return TA[kg]; },
                    [&](int mt, const f32x16& v, int s) { grad_epi(TB, GRD_ROW_T1, true, mt, v, s); });        // TB = dY_T1
            }
            // ---- albedo head backwards: albedo_pre -> A1 ----
            const U u_al = small_unit<P>(dalb, h);
            use2(pm8);
            run_layer<P, SLOT, 1, 4, false, NST>(ws, mid, lane, h, [&](int) { return u_al; },
                [&](int mt, const f32x16& v, int s) { grad_epi(DA1, GRD_ROW_A1, true, mt, v, s); });
            if constexpr (TRANS) {
                // ---- dY_T1 -> d embedding: the four embedding columns of the transient head's first layer (one m-tile, rows 0..3) ----
                run_layer<P, SLOT, QKG, 1, false>(ws, mid, lane, h, [&](int kg) { return TB[kg]; },
                    [&](int, const f32x16& v, int s) {
                        if (s == 0 && h == 0 && live) *reinterpret_cast<f32x4*>(a.g_emb + 4 * (size_t)p) = f32x4{v[0], v[1], v[2], v[3]};
                    });
            }
            // ---- [dY_A1, dY_T1 (transient head in the graph), d sigma_pre] -> dX8 -> mask(layer 7) -> dY7, through the heads' first
            //      layers folded with the bottleneck layer ----
            constexpr int AKG = TRANS ? 2 * QKG : QKG;
            const U u_sg = small_unit<P>(dsig, h);
            mb[0] = pm7[0]; mb[1] = pm7[1]; mb[2] = pm7[2]; mb[3] = pm7[3];
            auto src = [&](int kg) {
                if (kg < QKG) return DA1[kg < QKG ? kg : 0];
                if constexpr (TRANS) { if (kg < AKG) return TB[(kg >= QKG && kg < AKG) ? kg - QKG : 0]; }
                return u_sg;
            };
            if constexpr (PIPE == 1) {
                run_layer<P, SLOT, AKG + 1, 8, false>(ws, mid, lane, h, src,
                    [&](int mt, const f32x16& v, int s) { put_slice(P(), D, mt, s, mask_slice(P(), v, s, mt, mb[mt >> 1])); });
                // this wave's 32 samples = step wt of the pipeline: 16 units of 1 KiB
                uint8_t* dst = a.dy7_units + (size_t)wt * 16 * 1024 + lane * 16;
#pragma unroll
                for (int kg = 0; kg < HKG; ++kg) {
#if EO_DY7_NT
                    __builtin_nontemporal_store(D[kg], reinterpret_cast<U*>(dst + kg * 1024));      // written once, read once by the next launch
#else
                    *reinterpret_cast<U*>(dst + kg * 1024) = D[kg];
#endif
                }
                sw.drain();
                continue;
            }
            run_layer<P, SLOT, AKG + 1, 8, false, NST>(ws, mid, lane, h, src,
                [&](int mt, const f32x16& v, int s) { grad_epi(D, GRD_ROW_Y0 + 7 * 256, true, mt, v, s); });
        } else {
            const U u_sg = small_unit<P>(dsig, h);
            load_mask(7, 4);
            if constexpr (PIPE == 1) {      // density pass: dY_7 = W_sigma^T d sigma_pre .* relu'; trunk and input gradient follow elsewhere
                run_layer<P, SLOT, 1, 8, false>(ws, mid, lane, h, [&](int) { return u_sg; },
                    [&](int mt, const f32x16& v, int s) { put_slice(P(), D, mt, s, mask_slice(P(), v, s, mt, mb[mt >> 1])); });
                uint8_t* dst = a.dy7_units + (size_t)wt * 16 * 1024 + lane * 16;
#pragma unroll
                for (int kg = 0; kg < HKG; ++kg) *reinterpret_cast<U*>(dst + kg * 1024) = D[kg];
                sw.drain();
                continue;
            }
            run_layer<P, SLOT, 1, 8, false, NST>(ws, mid, lane, h, [&](int) { return u_sg; },
                [&](int mt, const f32x16& v, int s) { grad_epi(D, GRD_ROW_Y0 + 7 * 256, true, mt, v, s); });
        }

        // ---------------- trunk backwards: dY_l -> dX_l -> mask(layer l-1) -> dY_{l-1} ----------------
        f32x16 denc[2];
        auto trunk_step = [&](U* src, U* dst, int l) {      // consumes dY_l, produces dY_{l-1}
            load_mask(l - 1, 4);
            run_layer<P, SLOT, HKG, 8, false, NST>(ws, mid, lane, h, [&](int kg) { return src[kg]; },
                [&](int mt, const f32x16& v, int s) { grad_epi(dst, GRD_ROW_Y0 + (l - 1) * 256, true, mt, v, s); });
        };
        trunk_step(D, N, 7);
        trunk_step(N, D, 6);
        // layer 5 consumed [h, enc]: rows 0..255 go on down the trunk, rows 256..319 are d enc (skip path)
        load_mask(4, 4);
        run_layer<P, SLOT, HKG, IG ? 10 : 8, false>(ws, mid, lane, h, [&](int kg) { return D[kg]; },
            [&](int mt, const f32x16& v, int s) {
                if (mt < 8) { grad_epi(N, GRD_ROW_Y0 + 4 * 256, true, mt, v, s); return; }
                if constexpr (IG) { if (s == 0) denc[mt == 8 ? 0 : 1] = v; }
            });
        trunk_step(N, D, 4);
        trunk_step(D, N, 3);
        trunk_step(N, D, 2);
        trunk_step(D, N, 1);                                        // N = dY_0

        if constexpr (IG) {
            run_layer<P, SLOT, HKG, 2, false>(ws, mid, lane, h, [&](int kg) { return N[kg]; },
                [&](int mt, const f32x16& v, int s) {
                    if (s != 0) return;
#pragma unroll
                    for (int r = 0; r < 16; ++r) denc[mt == 0 ? 0 : 1][r] += v[r];
                });
            // encoder derivative: slot q = 16*t + r of lane half h (see encode_position / enc_col_of_hq)
            const float x = live ? a.px[p] : 0.f, y = live ? a.py[p] : 0.f, z = live ? a.pz[p] : 0.f;
            const float off = h ? EO_PI_2_F : 0.0f;
            float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = 16 * t + r;
                    const float g = denc[t][r];
                    if (q < 30) {
                        const int k = q / 3, d = q % 3;
                        const float cc = d == 0 ? x : (d == 1 ? y : z);
                        const float arg = cc * (float)(1 << k) + off;
                        const float dv = (P::IS_BF16 ? __cosf(arg) : cosf(arg)) * (float)(1 << k);
                        gp[d] += g * dv;
                    } else if (q == 30) {
                        if (h) gp[2] += g; else gp[0] += g;
                    } else {
                        if (!h) gp[1] += g;
                    }
                }
#pragma unroll
            for (int d = 0; d < 3; ++d) gp[d] += __shfl_xor(gp[d], 32, 64);       // both halves hold slots of the same sample
            if (h == 0 && live) {
                a.g_pos[p] = gp[0]; a.g_pos[(size_t)a.p_pad + p] = gp[1]; a.g_pos[2 * (size_t)a.p_pad + p] = gp[2];
            }
        }
        sw.drain();
    }
#ifdef EO_STAMP
    if (lane == 0) {
        atomicAdd(&eo_stamps_bwd[0], EO_T() - t_begin); atomicAdd(&eo_stamps_bwd[1], ws.t_wait); atomicAdd(&eo_stamps_bwd[2], ws.t_bar); atomicAdd(&eo_stamps_bwd[3], ws.t_flush); atomicAdd(&eo_stamps_bwd[4], 1ull);
    }
#endif
}

template <class P, bool FULL, bool IG, bool TRANS, int PIPE = 0>
hipError_t launch(const MlpBwdArgs& a, int grid, hipStream_t st) {
    constexpr int SMEM = 2 * FwdSlot<P>::BYTES + SlabWriter<P, GrdMap>::LDS_BYTES;
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd<P, FULL, IG, TRANS, PIPE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_mlp_bwd<P, FULL, IG, TRANS, PIPE>), dim3(grid), dim3(P::NT), SMEM, st, a);
    return hipGetLastError();
}

template <class P> hipError_t dispatch(const MlpBwdArgs& a, bool full, bool input_grad, bool transient, int grid, hipStream_t st) {
    if (!full) return launch<P, false, true, false>(a, grid, st);
    if (input_grad) return launch<P, true, true, true>(a, grid, st);
    return transient ? launch<P, true, false, true>(a, grid, st) : launch<P, true, false, false>(a, grid, st);
}

}  // namespace

// the variants the callers need: camera pass (all heads, with / without the transient head in the graph, no input grad),
// shadow pass / query_density (density only, with input grad) and the differentiable EONerfMLP.forward (all heads + input grad)
hipError_t eo_launch_mlp_bwd(const MlpBwdArgs& a, bool bf16, bool full, bool input_grad, bool transient, int grid, hipStream_t st, int pipe) {
    if (!full && !input_grad) return hipErrorInvalidValue;
    if (full && input_grad && !transient) return hipErrorInvalidValue;
    if (pipe) {      // heads only; the trunk runs in eonerf_bwd_pipe.hip (bf16), the input gradient in eonerf_ig_tail.hip
        if (!bf16 || !a.dy7_units || (full && input_grad)) return hipErrorInvalidValue;
        if (!full) return launch<PBf16, false, true, false, 1>(a, grid, st);
        return transient ? launch<PBf16, true, false, true, 1>(a, grid, st) : launch<PBf16, true, false, false, 1>(a, grid, st);
    }
    return bf16 ? dispatch<PBf16>(a, full, input_grad, transient, grid, st) : dispatch<PF32>(a, full, input_grad, transient, grid, st);
}

#ifdef EO_STAMP
extern "C" void eonerf_debug_read_bwd(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(eo_stamps_bwd), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(eo_stamps_bwd), z, sizeof(z));
}
#endif
