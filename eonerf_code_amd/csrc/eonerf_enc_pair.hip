// The shadow pass' two products against the encoding AND its input-gradient tail in ONE pass over dY_0 / dY_5 (round 6, VERDICT r5 #5b).
//
// Behind the shadow pass' pipelined trunk launch (eonerf_bwd_pipe.hip) two consumers read the tiles dY_0 and dY_5 its stages of layers 1
// and 6 left in the gradient slab (B-operand unit order, 16 KiB per 32 samples and tensor):
//   * eonerf_ig_tail.hip:   d enc = W_0^T dY_0 + W_5[:, 256:]^T dY_5  ->  encoder derivative  ->  d sigma / d position   (sat_rendering.py:90
//                           keeps the rendered depth attached: this gradient flows on into the camera pass, so it is needed at once)
//   * eonerf_wgrad.hip:     dW_0 += dY_0 enc^T,  dW_5[:, 256:] += dY_5 enc^T,  db_0 += row sums of dY_0   (two jobs of the GEMM launch at the
//                           END of the backward: the tiles come back from HBM a second time, 0.36 GB per step)
// Here one workgroup streams its share of the sample tiles ONCE -- per 32-sample step dY_0 (16 KiB), dY_5 (16 KiB) and the 64 encoding rows
// (4 KiB) by LDS-DMA into a 4-slot ring, three steps in flight -- and does both:
//   waves 0-3: d enc.  Wave (s, mt) multiplies m-tile mt of W_s^T (16 A units, stationary in registers) with the tile of source s (the B units as
//              they lie: 16 MFMAs), applies the encoder derivative to ITS 16 slots x 2 halves (the derivative is linear in d enc) and leaves
//              three partial sums per sample in LDS; one step later (behind the next step's barrier) wave 0 adds the four partials and stores
//              d sigma / d position.  These waves issue no LDS-DMA: their position loads (prefetched a step ahead) are the only vector-memory
//              operations the compiler has to wait for -- a wave with DMA in flight would drain its whole prefetch for them (vmcnt is in order).
//   waves 4-7: the staging (9 pieces of 1 KiB per wave and step) and the weight gradients.  Wave (s, half) owns four m-tiles x both n-tiles of
//              dW_s (128 accumulator registers, stationary for the launch): A fragments = transposed reads of the unit-order tile (as
//              eonerf_wgrad.hip's a_units jobs), B fragments = the encoding rows (swizzled like the GEMM's operand tiles): 16 MFMAs per step;
//              the bias gradient db_0 from the A fragments (as eonerf_bwd_pipe.hip).
//   End of the launch: every workgroup stores its 2 x 256 x 64 accumulators as a partial (EncPairArgs::part); k_step_tail sums the partials into
//   the gradient buffer at the end of the backward (an atomic flush of 256 workgroups into the same 64 + 80 KB took longer than the streaming).
// One barrier per step.  Not used in deterministic mode (EONERF_DETERMINISTIC: the fixed-order reductions stay with ig_tail + GEMM jobs).
#include "eonerf_common.h"
#include "eonerf_kernels.h"

// Diagnostic builds only (scripts/enc_pair_ablate.sh): EO_EP_ABL bit 0 (the atomic flush of the first version) is gone, bit 1 the d enc waves' work (MFMAs + encoder
// derivative), bit 2 the dW waves' MFMAs and fragment reads, bit 3 the LDS-DMA refill behind the prologue.  Results are WRONG with any bit set.
#ifndef EO_EP_ABL
#define EO_EP_ABL 0
#endif

namespace {

constexpr int NT = 512;
constexpr int IMG = 16 * 1024;                 // one sample tile of a 256-row block in unit order
constexpr int ENC_B = ENC_SLOTS * SEG_B;       // 4 KiB: the 64 encoding rows of one sample tile
constexpr int SLOT_B = 2 * IMG + ENC_B;        // dY_0 | dY_5 | enc
constexpr int NS = 4, DEPTH = NS - 1;
constexpr int N_DMA = 9;                       // pieces per staging wave and step: 4 + 4 units of the two tiles, 16 encoding rows
constexpr int GP_F = 4 * 32 * 4;               // floats per parity: [wave 4][sample 32][x y z pad]
constexpr int SMEM_B = NS * SLOT_B + 2 * GP_F * (int)sizeof(float);

EO_DEV int swz16(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 16; }      // eonerf_wgrad.hip's operand swizzle

// ---- waves 0-3 -------------------------------------------------------------------------------------------------------------------
EO_DEV void enc_waves(const EncPairArgs& a, uint8_t* smem, int lane, int wave, int t0, int t1, int n_pts) {
    typedef PBf16 P;
    typedef P::U U;
    const int h = lane >> 5, c = lane & 31;
    const int s_src = wave >> 1, mt = wave & 1;      // wave-uniform
    U wt[16];
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) wt[kg] = *reinterpret_cast<const U*>(a.wt + ((size_t)((s_src * 2 + mt) * 16 + kg)) * 1024 + lane * 16);
    float* gp_lds = reinterpret_cast<float*>(smem + NS * SLOT_B);
    const float eoff = h ? EO_PI_2_F : 0.0f;
    auto load_pos = [&](int t, float& x, float& y, float& z) {
        const int p = t * 32 + c;
        const bool live = p < n_pts;
        x = live ? a.px[p] : 0.f; y = live ? a.py[p] : 0.f; z = live ? a.pz[p] : 0.f;
    };
    // wave 0, one step late: the four partial sums of the previous step -> d sigma / d position
    auto finish = [&](int t_prev, int par) {
        const int p = t_prev * 32 + c;
        if (h == 0 && p < n_pts) {
            const f32x4* g = reinterpret_cast<const f32x4*>(gp_lds + par * GP_F);
            f32x4 v = g[c];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const f32x4 u = g[w * 32 + c]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; }
            a.g_pos[p] = v[0]; a.g_pos[(size_t)a.p_pad + p] = v[1]; a.g_pos[2 * (size_t)a.p_pad + p] = v[2];
        }
    };
    float x, y, z;
    load_pos(t0, x, y, z);
    for (int t = t0; t < t1; ++t) {
        const int slot = (t - t0) & (NS - 1), par = (t - t0) & 1;
        asm volatile("s_barrier" ::: "memory");
        if (wave == 0 && t > t0) finish(t - 1, par ^ 1);
        if (EO_EP_ABL & 2) continue;
        // ---- d enc partial of (source s_src, m-tile mt): 16 MFMAs over the tile's 16 B units ----
        const uint8_t* bp = smem + slot * SLOT_B + s_src * IMG + lane * 16;
        f32x16 acc = zero_acc();
        U fr[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) fr[d] = lds_unit<P>(bp + d * 1024);
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) {
            acc = P::mma(wt[kg], fr[kg % 3], acc);
            if (kg + 3 < 16) fr[kg % 3] = lds_unit<P>(bp + (kg + 3) * 1024);
        }
        // encoder derivative of this wave's 16 slots (eonerf_ig_tail.hip: slot q = 16 mt + r of lane half h)
        float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = acc[r];
            if (mt == 0 || r < 14) {      // slots q < 30: sin / cos terms of frequency 2^(q / 3), coordinate q % 3
                const int q0 = r, q1 = 16 + r;
                const int k = mt ? q1 / 3 : q0 / 3, d = mt ? q1 % 3 : q0 % 3;      // (mt is wave-uniform; r is a compile-time constant)
                const float cc = d == 0 ? x : (d == 1 ? y : z);
                const float arg = cc * (float)(1 << k) + eoff;
                gp[d] += g * (__cosf(arg) * (float)(1 << k));
            } else if (r == 14) {         // q = 30: the identity terms x (half 0) / z (half 1)
                if (h) gp[2] += g; else gp[0] += g;
            } else {                      // q = 31: y (half 0) / padding
                if (!h) gp[1] += g;
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) gp[d] += __shfl_xor(gp[d], 32, 64);
        if (h == 0) reinterpret_cast<f32x4*>(gp_lds + par * GP_F)[wave * 32 + c] = f32x4{gp[0], gp[1], gp[2], 0.f};
        if (t + 1 < t1) load_pos(t + 1, x, y, z);      // a whole step ahead of its use
    }
    asm volatile("s_barrier" ::: "memory");
    if (wave == 0) finish(t1 - 1, (t1 - 1 - t0) & 1);
}

// ---- waves 4-7 -------------------------------------------------------------------------------------------------------------------
EO_DEV void dw_waves(const EncPairArgs& a, uint8_t* smem, int lane, int wave, int t0, int t1) {
    typedef PBf16 P;
    typedef P::U U;
    const int h = lane >> 5, c = lane & 31, v = wave - 4;
    const int s_src = v >> 1, mhalf = v & 1;      // wave-uniform: source (0: dY_0 / layer 0, 1: dY_5 / skip columns of layer 5), m-tiles 4 mhalf ..
    const size_t nt = (size_t)a.p_pad / 32;
    const uint8_t* grd = reinterpret_cast<const uint8_t*>(a.grd);
    const __amdgpu_buffer_rsrc_t rs_y0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(grd) + (size_t)GRD_ROW_Y0 * nt * SEG_B, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y5 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(grd) + (size_t)(GRD_ROW_Y0 + 5 * 256) * nt * SEG_B, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(a.act)) + (size_t)ACT_ROW_ENC * nt * SEG_B, 0, -1, 0x00020000);
    // staging: wave v copies units 4v .. 4v+3 of both tiles and 16 encoding rows (source chunks XOR-swizzled so that a B-fragment read of
    // 16 rows x one chunk covers all banks once)
    const int enc_row = 16 * v + (lane >> 2);
    const int enc_voff = enc_row * SEG_B + (((lane & 3) ^ ((enc_row >> 2) & 3)) * 16);
    auto issue = [&](int t, int slot) {
        uint8_t* base = smem + slot * SLOT_B;
        const uint32_t so = (uint32_t)t * IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y0, (__attribute__((address_space(3))) void*)(base + (4 * v + i) * 1024), 16, lane * 16 + (4 * v + i) * 1024, so, 0, 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y5, (__attribute__((address_space(3))) void*)(base + IMG + (4 * v + i) * 1024), 16, lane * 16 + (4 * v + i) * 1024, so, 0, 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_e, (__attribute__((address_space(3))) void*)(base + 2 * IMG + 16 * v * SEG_B), 16, enc_voff, (uint32_t)t * ENC_B, 0, 2);
    };
    f32x16 dw[4][2];
    float dbv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { dw[i][0] = zero_acc(); dw[i][1] = zero_acc(); }
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    int off_a[4];                // transposed A-fragment reads of this wave's four m-tiles (+ 64: second half, + 256: second K step)
#pragma unroll
    for (int i = 0; i < 4; ++i)
        off_a[i] = s_src * IMG + (2 * (4 * mhalf + i) + (g4 & 1)) * 1024 + ((pp & 1) * 32 + 8 * (g4 >> 1) + qq) * 16 + (pp >> 1) * 8;
    int off_b[2][2];             // B fragments: encoding row 32 j + c, chunk 2 ks + h
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int row = 32 * j + c; off_b[j][ks] = 2 * IMG + row * SEG_B + swz16(row, 2 * ks + h); }

#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(t0 + d < t1 ? t0 + d : t1 - 1, d);
    for (int t = t0; t < t1; ++t) {
        const int slot = (t - t0) & (NS - 1);
        // this wave's share of step t has landed once at most DEPTH - 1 younger steps are outstanding; the barrier publishes every wave's
        // share and retires all reads of the slot refilled next
        if (EO_EP_ABL & 8) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N_DMA * (DEPTH - 1)) : "memory");
        if (!(EO_EP_ABL & 8)) issue(t + DEPTH < t1 ? t + DEPTH : t1 - 1, (slot + DEPTH) & (NS - 1));
        const uint8_t* T = smem + slot * SLOT_B;
        if (EO_EP_ABL & 4) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (inline asm: for the intrinsic the wait-count pass assumes aliasing with the LDS-DMA in flight and drains it)
            u32x2 ta[8];
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                         "ds_read_b64_tr_b16 %1, %8 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %9\n\t"
                         "ds_read_b64_tr_b16 %3, %9 offset:64\n\t"
                         "ds_read_b64_tr_b16 %4, %10\n\t"
                         "ds_read_b64_tr_b16 %5, %10 offset:64\n\t"
                         "ds_read_b64_tr_b16 %6, %11\n\t"
                         "ds_read_b64_tr_b16 %7, %11 offset:64\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]), "=&v"(ta[4]), "=&v"(ta[5]), "=&v"(ta[6]), "=&v"(ta[7])
                         : "v"((uint32_t)(uintptr_t)(T + off_a[0] + 256 * ks)), "v"((uint32_t)(uintptr_t)(T + off_a[1] + 256 * ks)),
                           "v"((uint32_t)(uintptr_t)(T + off_a[2] + 256 * ks)), "v"((uint32_t)(uintptr_t)(T + off_a[3] + 256 * ks)) : "memory");
            const U b0 = lds_unit<P>(T + off_b[0][ks]), b1 = lds_unit<P>(T + off_b[1][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 av = u32x4{ta[2 * i][0], ta[2 * i][1], ta[2 * i + 1][0], ta[2 * i + 1][1]};
                const U af = __builtin_bit_cast(U, av);
                dw[i][0] = P::mma(af, b0, dw[i][0]);
                dw[i][1] = P::mma(af, b1, dw[i][1]);
                if (s_src == 0) {      // bias gradient of layer 0: this lane's 8 samples of feature row 32 (4 mhalf + i) + c
#pragma unroll
                    for (int e = 0; e < 4; ++e) dbv[i] += __uint_as_float(av[e] << 16) + __uint_as_float(av[e] & 0xffff0000u);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the clamped tail prefetches)
    // ---- this workgroup's partial: [source][row][slot], one 128-B segment per half-wave instruction; db_0 behind it ----
    float* part = a.part + (size_t)blockIdx.x * ENC_PART_F + (size_t)s_src * 256 * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) part[((4 * mhalf + i) * 32 + acc_row(g, h)) * 64 + 32 * j + c] = dw[i][j][g];
    if (s_src == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sum = dbv[i] + __shfl_xor(dbv[i], 32, 64);
            if (h == 0) a.part[(size_t)blockIdx.x * ENC_PART_F + 2 * 256 * 64 + (4 * mhalf + i) * 32 + c] = sum;
        }
    }
}

__global__ __launch_bounds__(NT) void k_enc_pair(EncPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_pts = *a.n_pts;
    const int n_tiles = (n_pts + 255) / 256 * 8;      // whole 256-sample tiles, as the pipelined launch wrote them (dead samples: zero gradients)
    const int t0 = (int)((long long)blockIdx.x * n_tiles / gridDim.x), t1 = (int)((long long)(blockIdx.x + 1) * n_tiles / gridDim.x);
    if (t0 >= t1) {      // (uniform per workgroup; fewer tiles than workgroups) an all-zero partial
        for (int i = threadIdx.x; i < ENC_PART_F; i += NT) a.part[(size_t)blockIdx.x * ENC_PART_F + i] = 0.f;
        return;
    }
    if (wave < 4) enc_waves(a, smem, lane, wave, t0, t1, n_pts);
    else dw_waves(a, smem, lane, wave, t0, t1);
}

}  // namespace

hipError_t eo_launch_enc_pair(const EncPairArgs& a, int n_wg, hipStream_t st) {
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_pair), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_enc_pair, dim3(n_wg), dim3(NT), SMEM_B, st, a);
    return hipGetLastError();
}
