// Weight-gradient GEMMs of the EO-NeRF field (the dW half of H10):
//     dW[m][n] += sum_p dY^T[m][p] * X^T[n][p]          db[m] += sum_p dY^T[m][p]
// Both operands are the feature-major slabs the chain kernels saved (rows = features, samples contiguous), so
// this is an "NT" GEMM whose contraction runs over the ~5e5 samples of a batch and whose output is tiny: every
// workgroup owns a whole (<= 256 x 256) output tile in registers for a slice of the sample range (split-K over
// workgroups, all layers of both passes in ONE launch) and flushes it with fp32 atomics at the end.
// The kernel is HBM-bound (131 FLOP/B: ~650 TFLOP/s at 5 TB/s), so the loop is built to keep bytes in flight: a 4-slot LDS
// ring filled by LDS-DMA (buffer_load_dwordx4 ... lds, no staging registers), three K steps (96 KiB) outstanding per CU
// behind a COUNTED s_waitcnt vmcnt, one raw s_barrier per step.  LDS tiles are [rows][64 B] (bf16: 32 samples, fp32:
// 16), 16-B chunks XOR-swizzled through the per-lane SOURCE address so the ds_read_b128 operand reads are
// bank-conflict free.
#include "eonerf_wgrad_dev.h"

namespace {
using namespace eo_wgrad;

template <class P>
__global__ __launch_bounds__(WG_NT) void k_wgrad(const WgradJobTable tab, int p_pad, int* queue, float* partials) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    (void)p_pad;
    wgrad_work<P, WgradJobTable, false>(tab, queue, partials, smem, 0, nullptr);
}

// deterministic mode: block = row, thread = column; the jobs one after the other in table order, a job's non-empty slices in slice
// order.  (Round 5: the jobs used to be a second grid dimension, "+=" without atomics -- but the camera pass' and the shadow pass' jobs of
// one layer (layer 0, the skip columns, the sigma row) add into the SAME elements: two blocks could read-modify-write one address at the same
// time and lose a contribution, once in a few hundred steps.  One block per row walks every job, so an element has one writer.)
template <class P>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgradJobTable tab, const float* partials) {
    const int row = blockIdx.x, col = threadIdx.x;
    for (int jb = 0; jb < tab.n; ++jb) {
        const WgradJob job = tab.j[jb];
        if (row >= job.m_rows) continue;
        const int n_pts = *job.n_pts;
        const int steps = (n_pts + P::TILE - 1) / P::TILE * P::TILE / (ROW_B / P::ACT_BYTES);
        float acc = 0.f, accb = 0.f;
        for (int w = 0; w < job.slices; ++w) {
            const int s0 = (int)((long long)w * steps / job.slices), s1 = (int)((long long)(w + 1) * steps / job.slices);
            if (s0 >= s1) continue;
            const float* pt = partials + (size_t)(job.item0 + w) * WGRAD_PART_F;
            if (col < job.n_rows) acc += pt[row * 256 + col];
            if (col == 0 && job.db) accb += pt[256 * 256 + row];
        }
        const bool second = row >= job.split;
        if (col < job.n_rows) {
            const int cm = job.col_map ? job.col_map[col] : col;
            if (cm >= 0 && (second || job.dw)) (second ? job.dw2 + (size_t)(row - job.split) * job.dw2_ld : job.dw + (size_t)row * job.dw_ld)[cm] += acc;
        }
        if (col == 0 && job.db) (second ? job.db2 + (row - job.split) : job.db + row)[0] += accb;
    }
}

template <class P> hipError_t launch(const WgradJobTable& jobs, int n_wg, int p_pad, int* queue, hipStream_t st, float* partials) {
    constexpr int SMEM = WG_SMEM;
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<P>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_wgrad<P>), dim3(n_wg), dim3(WG_NT), SMEM, st, jobs, p_pad, queue, partials);
    if (partials) hipLaunchKernelGGL((k_wgrad_reduce<P>), dim3(256), dim3(256), 0, st, jobs, partials);
    return hipGetLastError();
}

}  // namespace

hipError_t eo_launch_wgrad(const WgradJobTable& jobs, int n_wg, int p_pad, int* queue, bool bf16, hipStream_t st, float* partials, bool zero_queue) {
    if (zero_queue) {
        hipError_t e = hipMemsetAsync(queue, 0, sizeof(int), st);
        if (e != hipSuccess) return e;
    }
    return bf16 ? launch<PBf16>(jobs, n_wg, p_pad, queue, st, partials) : launch<PF32>(jobs, n_wg, p_pad, queue, st, partials);
}
