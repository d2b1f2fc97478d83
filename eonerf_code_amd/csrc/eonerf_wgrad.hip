// Weight-gradient GEMMs of the EO-NeRF field (the dW half of H10):
//     dW[m][n] += sum_p dY^T[m][p] * X^T[n][p]          db[m] += sum_p dY^T[m][p]
// Both operands are the feature-major slabs the chain kernels saved (rows = features, samples contiguous), so
// this is an "NT" GEMM whose contraction runs over the ~5e5 samples of a batch and whose output is tiny: every
// workgroup owns a whole (<= 256 x 256) output tile in registers for a slice of the sample range (split-K over
// workgroups, all layers of both passes in ONE launch) and flushes it with fp32 atomics at the end.
// LDS tiles: [rows][128 B] (bf16: 64 samples, fp32: 32 samples), 16-B chunks XOR-swizzled by (row>>1)&7 so that the
// ds_read_b128 operand reads (16 different rows per lane group) are bank-conflict free; register-staged double
// buffering (global -> VGPR early, VGPR -> LDS after the MFMAs, one barrier per K step).
#include "eonerf_common.h"
#include "eonerf_kernels.h"

namespace {

constexpr int WG_NT = 512;
constexpr int MAX_ROWS = 256;
constexpr int ROW_B = 128;                       // bytes per LDS row
constexpr int TILE_B = MAX_ROWS * ROW_B;         // 32 KiB per operand buffer
constexpr int WMAX = 4, NMAX = 2;

EO_DEV int swz(int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) * 16; }

template <class P>
__global__ __launch_bounds__(WG_NT) void k_wgrad(const WgradJobTable tab, int p_pad) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef typename P::U U;
    constexpr int BK = ROW_B / P::ACT_BYTES;             // samples per K step
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, r = lane & 31;

    int ji = 0;
    for (int j = 0; j < tab.n; ++j) if ((int)blockIdx.x >= tab.j[j].wg_begin) ji = j;
    const WgradJob job = tab.j[ji];
    const int w = blockIdx.x - job.wg_begin;
    const int n_pts = *job.n_pts;
    const int n_pad = (n_pts + P::TILE - 1) / P::TILE * P::TILE;
    const int steps = n_pad / BK;
    const int s0 = (int)((long long)w * steps / job.wg_count), s1 = (int)((long long)(w + 1) * steps / job.wg_count);
    if (s0 >= s1) return;

    const int wm = job.wm, wn = job.wn, gn = job.gn;
    const int a_rows = job.gm * wm * 32, b_rows = gn * wn * 32;
    const bool active = wid < job.gm * gn;
    const int wm_idx = wid / gn, wn_idx = wid % gn;

    uint8_t* lds_a = smem;                      // [2][TILE_B]
    uint8_t* lds_b = smem + 2 * TILE_B;         // [2][TILE_B]

    // ---- staging: thread -> (row = tid/8 + 64*pass, 16-B chunk = tid%8) ----
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    u32x4 ra[4], rb[4];
    auto fetch = [&](int step) {
        const size_t k_off = (size_t)step * ROW_B + ld_chunk * 16;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ld_row + 64 * ps;
            ra[ps] = u32x4{0, 0, 0, 0};
            rb[ps] = u32x4{0, 0, 0, 0};
            if (row < a_rows && row < job.m_rows)
                ra[ps] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(job.a) + (size_t)row * p_pad * P::ACT_BYTES + k_off);
            if (row < b_rows && row < job.n_rows)
                rb[ps] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(job.b) + (size_t)row * p_pad * P::ACT_BYTES + k_off);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ld_row + 64 * ps;
            if (row < a_rows) *reinterpret_cast<u32x4*>(lds_a + buf * TILE_B + row * ROW_B + swz(row, ld_chunk)) = ra[ps];
            if (row < b_rows) *reinterpret_cast<u32x4*>(lds_b + buf * TILE_B + row * ROW_B + swz(row, ld_chunk)) = rb[ps];
        }
    };

    f32x16 acc[WMAX][NMAX];
    f32x16 accb = zero_acc();
#pragma unroll
    for (int i = 0; i < WMAX; ++i)
#pragma unroll
        for (int j = 0; j < NMAX; ++j) acc[i][j] = zero_acc();
    U ones = P::zero();
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < P::NE; ++e) { if constexpr (P::IS_BF16) ones[e] = (__bf16)1.0f; else ones[e] = 1.0f; }
    }
    const bool do_bias = job.db != nullptr && active && wn_idx < wm;

    fetch(s0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int s = s0; s < s1; ++s) {
        const bool more = s + 1 < s1;
        if (more) fetch(s + 1);
        if (active) {
            const uint8_t* A = lds_a + cur * TILE_B;
            const uint8_t* B = lds_b + cur * TILE_B;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int chunk = 2 * kg + h;
                U af[WMAX], bf[NMAX];
#pragma unroll
                for (int i = 0; i < WMAX; ++i)
                    if (i < wm) { const int row = (wm_idx * wm + i) * 32 + r; af[i] = lds_unit<P>(A + row * ROW_B + swz(row, chunk)); }
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < wn) { const int row = (wn_idx * wn + j) * 32 + r; bf[j] = lds_unit<P>(B + row * ROW_B + swz(row, chunk)); }
#pragma unroll
                for (int i = 0; i < WMAX; ++i)
                    if (i < wm) {
#pragma unroll
                        for (int j = 0; j < NMAX; ++j)
                            if (j < wn) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
                        if (do_bias && i == wn_idx) accb = P::mma(af[i], ones, accb);
                    }
            }
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- flush: fp32 atomics, 32 consecutive columns per half-wave instruction ----
    if (!active) return;
#pragma unroll
    for (int i = 0; i < WMAX; ++i)
        if (i < wm) {
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (j < wn) {
                    const int col = (wn_idx * wn + j) * 32 + r;
                    int cm = col;
                    if (col < job.n_rows && job.col_map) cm = job.col_map[col];
                    if (col < job.n_rows && cm >= 0) {
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const int row = (wm_idx * wm + i) * 32 + acc_row(g, h);
                            if (row < job.m_rows) atomicAdd(job.dw + (size_t)row * job.dw_ld + cm, acc[i][j][g]);
                        }
                    }
                }
        }
    if (do_bias && r == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = (wm_idx * wm + wn_idx) * 32 + acc_row(g, h);
            if (row < job.m_rows) atomicAdd(job.db + row, accb[g]);
        }
    }
}

template <class P> hipError_t launch(const WgradJobTable& jobs, int n_wg, int p_pad, hipStream_t st) {
    constexpr int SMEM = 4 * TILE_B;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<P>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_wgrad<P>), dim3(n_wg), dim3(WG_NT), SMEM, st, jobs, p_pad);
    return hipGetLastError();
}

}  // namespace

hipError_t eo_launch_wgrad(const WgradJobTable& jobs, int n_wg, int p_pad, bool bf16, hipStream_t st) {
    return bf16 ? launch<PBf16>(jobs, n_wg, p_pad, st) : launch<PF32>(jobs, n_wg, p_pad, st);
}
