// Workspace carving of the C ABI (eonerf_render_workspace_bytes / eonerf_render_forward / _backward; eonerf_field_*): plain host code,
// no HIP runtime calls -- shared by eonerf_api.hip and the host-only sanitizer test (tests/host/host_checks.cpp, built for the CPU
// under the address + undefined-behaviour sanitizers).  The caller (PyTorch) owns the workspace; these functions only lay it out.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "../../include/eonerf_hip.h"
#include "eonerf_kernels.h"
#include "eonerf_rays.h"

constexpr int BOTT_SCRATCH_F = 2 * 128 * 256 + 256;       // bottleneck factors M_a | M_t, then db_A1 | db_T1

struct CarveCfg {          // what the layout depends on besides (n_rays, flags): see eonerf_ctx
    bool bf16 = true, pipe = false, deterministic = false, pipe_partials = false;
    int n_pipes = 0;
    int enc_part_wgs = 0;  // > 0: workgroups of eonerf_enc_pair.hip, each with a partial of the shadow pass' encoding products (0: ig_tail + GEMM jobs)
    int n_samples = 128;   // int(2 / render_step_size) of the calls this layout serves (eonerf_set_n_samples)
};

// bump allocator over the caller's workspace (256-byte aligned); with base == nullptr it only measures
struct Carver {
    uint8_t* base; size_t off = 0;
    explicit Carver(void* b) : base(reinterpret_cast<uint8_t*>(b)) {}
    template <class T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct PassBuffers {       // one MLP pass (camera or sun) over up to p_cap samples
    int *counts, *offsets, *n_pts;
    float *px, *py, *pz, *tmid, *delta, *sigma, *albedo, *ts, *tb;
    int* simg;
    void *act, *grd; uint32_t* masks;
    float *g_sigma, *g_albedo, *g_ts, *g_tb, *g_emb, *g_pos;
};

struct PipeWs {            // layer-pipelined backward (eonerf_bwd_pipe.hip)
    uint8_t* dy_in;        // dY_7 in unit order: p_cap x 512 B (heads chain -> trunk pipeline)
    uint8_t* rings;        // [pipelines][edges][PIPE_RING][16 KiB]
    uint32_t* sync;        // PIPE_LAUNCHES consecutive blocks, one per pipelined launch of a backward call (sun trunk, camera trunk):
                           // [32] role counter, [64..) one scratch line per workgroup, then the edge flags -- all zeroed by ONE memset
    size_t sync_bytes;     // of one block
};
constexpr int PIPE_LAUNCHES = 2;

struct DetWs {             // EONERF_DETERMINISTIC: partial sums instead of atomics
    float* pipe_part;      // [n_pipes * 7][256 * 256 + 256]
    float* wgrad_part;     // [WGRAD_MAX_JOBS * 48 items][256 * 256 + 256]
    float* rad_rays;       // [R][6]
    float* emb_rays;       // [R][4]
};

struct RenderWs {
    PipeWs pipe;
    DetWs det;
    int *cnt_first, *cnt_retry, *flags;
    float* ray_rec; float* g_ray; float* amb_save;
    float* m_bott;        // [2][128][256] fp32: dA1^T X8 and dT1^T X8 (the bottleneck factors), then [256] db_A1 | db_T1 of this call
    int* queue;           // work-item counter of the weight-gradient GEMM (behind them)
    float* enc_part;      // [enc_part_wgs][ENC_PART_F] partials of eonerf_enc_pair.hip (training with the shadow pass on the pipelined path), or nullptr
    PassBuffers cam, sun;
    size_t bytes;
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// capacity of the compact per-sample arrays of a pass: a ray has n_samples - 1 intervals at most (n_samples = int(2 / render_step_size))
inline int p_cap_of(int n_rays, int n_samples) { return round_up(std::max(n_rays, 1) * (n_samples - 1), 256); }
// a 256-row block of a training slab ([sample tile][256 rows][64 B]) is addressed through ONE buffer descriptor with 32-bit byte offsets
// (SlabWriter, the GEMM's and the pipeline's operand loads): p_cap / samples-per-tile x 16 KiB must stay below 4 GiB --
// 66,050 rays (8.39 M samples) per call in bf16 mode, 33,024 in fp32 mode.  Larger batches are chunked by the caller (render_image does)
inline bool slab_blocks_addressable(bool bf16, size_t p_cap) { return p_cap / (bf16 ? 32 : 16) * 256 * SEG_B < (1ull << 32); }

inline void carve_pass(Carver& c, PassBuffers& b, int n_rays, int p_cap, bool full, bool train, bool input_grad, int act_bytes) {
    b.counts = c.take<int>(n_rays);
    b.offsets = c.take<int>(n_rays + 1);
    b.n_pts = c.take<int>(4);
    b.px = c.take<float>(p_cap); b.py = c.take<float>(p_cap); b.pz = c.take<float>(p_cap);
    b.tmid = c.take<float>(p_cap); b.delta = c.take<float>(p_cap);
    b.simg = c.take<int>(p_cap);
    b.sigma = c.take<float>(p_cap);
    b.albedo = full ? c.take<float>(3 * (size_t)p_cap) : nullptr;
    b.ts = full ? c.take<float>(p_cap) : nullptr;
    b.tb = full ? c.take<float>(p_cap) : nullptr;
    b.act = b.grd = nullptr; b.masks = nullptr;
    b.g_sigma = b.g_albedo = b.g_ts = b.g_tb = b.g_emb = b.g_pos = nullptr;
    if (train) {
        b.act = c.take<uint8_t>((size_t)(full ? ACT_ROWS_FULL : ACT_ROWS_DENSITY) * p_cap * act_bytes);
        b.grd = c.take<uint8_t>((size_t)(full ? GRD_ROWS_FULL : GRD_ROWS_DENSITY) * p_cap * act_bytes);
        b.masks = c.take<uint32_t>((size_t)(full ? MASK_SLOTS_FULL : MASK_SLOTS_DENSITY) * p_cap * 8);
        b.g_sigma = c.take<float>(p_cap);
        if (full) {
            b.g_albedo = c.take<float>(3 * (size_t)p_cap);
            b.g_ts = c.take<float>(p_cap); b.g_tb = c.take<float>(p_cap);
            b.g_emb = c.take<float>(4 * (size_t)p_cap);
        }
        if (input_grad) b.g_pos = c.take<float>(3 * (size_t)p_cap);
    }
}

inline RenderWs carve_render(const CarveCfg& cfg, void* base, int n_rays, int flags) {
    const CarveCfg* ctx = &cfg;
    Carver c(base);
    RenderWs w;
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);
    const bool train = flags & EONERF_F_TRAIN, shadows = flags & EONERF_F_SHADOWS, od = flags & EONERF_F_ONLY_DEPTH;
    const int ab = ctx->bf16 ? 2 : 4;
    w.cnt_first = c.take<int>(n_rays); w.cnt_retry = c.take<int>(n_rays); w.flags = c.take<int>(4);
    w.ray_rec = c.take<float>((size_t)n_rays * RAY_REC);
    w.g_ray = train ? c.take<float>((size_t)n_rays * RAY_REC) : nullptr;
    w.amb_save = train ? c.take<float>((size_t)n_rays * 160) : nullptr;
    // [bottleneck factors | GEMM work queue] and, right behind them, the pipeline's sync block: everything the backward needs zeroed, so
    // that the first pipeline launch of a backward call clears all of it with ONE memset
    w.m_bott = train ? c.take<float>(BOTT_SCRATCH_F + 64) : nullptr;
    w.queue = w.m_bott ? reinterpret_cast<int*>(w.m_bott + BOTT_SCRATCH_F) : nullptr;      // (measuring pass: no arithmetic on a null base)
    memset(&w.pipe, 0, sizeof(w.pipe));
    if (train && ctx->pipe) {
        const size_t wgs = (size_t)ctx->n_pipes * PIPE_STAGES;
        const size_t edges = (size_t)ctx->n_pipes * (PIPE_STAGES - 1);
        w.pipe.sync_bytes = (64 + wgs * 32 + edges * 64) * sizeof(uint32_t);
        w.pipe.sync = c.take<uint32_t>(PIPE_LAUNCHES * w.pipe.sync_bytes / sizeof(uint32_t));
        w.pipe.dy_in = c.take<uint8_t>((size_t)p_cap * 512);
        w.pipe.rings = c.take<uint8_t>(edges * PIPE_RING * PIPE_UNIT_B);
    }
    memset(&w.det, 0, sizeof(w.det));
    if (train && ctx->pipe && ctx->pipe_partials)
        w.det.pipe_part = c.take<float>((size_t)ctx->n_pipes * PIPE_STAGES * WGRAD_PART_F);
    if (train && ctx->deterministic) {
        w.det.wgrad_part = c.take<float>((size_t)WGRAD_MAX_JOBS * 48 * WGRAD_PART_F);
        w.det.rad_rays = c.take<float>((size_t)n_rays * 6);
        w.det.emb_rays = c.take<float>((size_t)n_rays * 4);
    }
    w.enc_part = (train && shadows && !od && ctx->pipe && ctx->enc_part_wgs > 0) ? c.take<float>((size_t)ctx->enc_part_wgs * ENC_PART_F) : nullptr;
    // (a density-only training pass -- render_depth under autograd -- may end in the chain kernel's input-gradient variant: room for d position)
    carve_pass(c, w.cam, n_rays, p_cap, !od, train, train && od, ab);
    if (shadows && !od) carve_pass(c, w.sun, n_rays, p_cap, false, train, true, ab); else memset(&w.sun, 0, sizeof(w.sun));
    w.bytes = c.off + 256;
    return w;
}

