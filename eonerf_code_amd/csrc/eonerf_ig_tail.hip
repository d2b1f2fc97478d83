// Input-gradient tail of the shadow pass when its trunk backward is layer-pipelined (eonerf_bwd_pipe.hip):
//     d enc = W_0^T dY_0  +  W_5[:, 256:]^T dY_5            (layer 0 and the skip columns of layer 5, radiance_fields/mlp.py:87-101)
//     d sigma / d position = sum over encoding slots of  d enc . d enc / d x      (encoder derivative 2^k cos(2^k x [+ pi/2]),
//                                                                                  radiance_fields/mlp.py:190-208)
// -- what the chain kernel's INPUT_GRAD variant computes at its end (eonerf_mlp_bwd.hip); sat_rendering.py:90 keeps the rendered
// depth attached, so this gradient flows on into the camera pass.  dY_0 and dY_5 are the tiles the pipeline's stages of layers 1
// and 6 wrote into the gradient slab in B-operand unit order: MFMA B operands as they are.
// Memory-bound (1 KiB per sample, 32 K MACs): one wave per 32-sample tile, 4 waves per workgroup, one per SIMD with the whole
// W^T operand (64 A units) in registers.
#include "eonerf_common.h"
#include "eonerf_kernels.h"

namespace {

constexpr int IMG = 16 * 1024;       // one sample tile of a 256-row block

__global__ __launch_bounds__(256) void k_ig_tail(IgTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef PBf16 P;
    typedef P::U U;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, c = lane & 31;
    uint8_t* img = smem + wave * 2 * IMG;
    const int n_pts = *a.n_pts;
    const int n_tiles_live = (n_pts + 31) / 32;
    const size_t nt = (size_t)a.p_pad / 32;
    const uint8_t* blk[2] = {reinterpret_cast<const uint8_t*>(a.grd) + (size_t)GRD_ROW_Y0 * nt * SEG_B,                  // dY_0
                             reinterpret_cast<const uint8_t*>(a.grd) + (size_t)(GRD_ROW_Y0 + 5 * 256) * nt * SEG_B};     // dY_5
    // stationary A operand: [source 2][m-tile 2][k-group 16] units
    U wt[2][2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kg = 0; kg < 16; ++kg) wt[s][mt][kg] = *reinterpret_cast<const U*>(a.wt + ((size_t)((s * 2 + mt) * 16 + kg)) * 1024 + lane * 16);
    for (int t = blockIdx.x * 4 + wave; t < n_tiles_live; t += gridDim.x * 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(blk[s]) + (size_t)t * 256 * SEG_B, 0, IMG, 0x00020000);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + s * IMG + j * 1024), 16, lane * 16, j * 1024, 0, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 acc[2] = {zero_acc(), zero_acc()};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int kg = 0; kg < 16; ++kg) {
                const U bu = lds_unit<P>(img + s * IMG + kg * 1024 + lane * 16);
                acc[0] = P::mma(wt[s][0][kg], bu, acc[0]);
                acc[1] = P::mma(wt[s][1][kg], bu, acc[1]);
            }
        }
        // encoder derivative (same arithmetic as the chain kernel's INPUT_GRAD tail): slot q = 16 t + r of lane half h
        const int p = t * 32 + c;
        const bool live = p < n_pts;
        const float x = live ? a.px[p] : 0.f, y = live ? a.py[p] : 0.f, z = live ? a.pz[p] : 0.f;
        const float off = h ? EO_PI_2_F : 0.0f;
        float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 16 * tt + r;
                const float g = acc[tt][r];
                if (q < 30) {
                    const int k = q / 3, d = q % 3;
                    const float cc = d == 0 ? x : (d == 1 ? y : z);
                    const float arg = cc * (float)(1 << k) + off;
                    gp[d] += g * (__cosf(arg) * (float)(1 << k));
                } else if (q == 30) {
                    if (h) gp[2] += g; else gp[0] += g;
                } else {
                    if (!h) gp[1] += g;
                }
            }
#pragma unroll
        for (int d = 0; d < 3; ++d) gp[d] += __shfl_xor(gp[d], 32, 64);
        if (h == 0 && live) {
            a.g_pos[p] = gp[0]; a.g_pos[(size_t)a.p_pad + p] = gp[1]; a.g_pos[2 * (size_t)a.p_pad + p] = gp[2];
        }
    }
}

}  // namespace

hipError_t eo_launch_ig_tail(const IgTailArgs& a, int n_wg, hipStream_t st) {
    constexpr int SMEM = 4 * 2 * IMG;
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ig_tail), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_ig_tail, dim3(n_wg), dim3(256), SMEM, st, a);
    return hipGetLastError();
}
