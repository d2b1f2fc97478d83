// C-ABI entry points of libeonerf_hip.so (include/eonerf_hip.h).  Host logic only: parameter layout, packed weight
// streams, workspace carving and kernel sequencing.  No device memory is allocated after eonerf_create.
#include <hip/hip_runtime.h>
#include <string.h>
#include <new>
#include <vector>
#include <unordered_map>

#include "../../include/eonerf_hip.h"
#include "eonerf_kernels.h"
#include "eonerf_pack.h"
#include "eonerf_rays.h"
#include "eonerf_raygen.h"
#include "eonerf_carve.h"
#include <math.h>
#include <stdio.h>
#ifdef EO_COR
hipError_t eo_launch_cor_partner(const void* src, size_t total_bytes, size_t stream_bytes, int n_wg, hipStream_t st);
#endif

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

struct DevStream {
    uint8_t* data = nullptr; size_t bytes = 0;
    ChunkDesc* chunks = nullptr; int n_chunks = 0;
    PackEntry *e16 = nullptr, *e32 = nullptr, *e16lo = nullptr; int n16 = 0, n32 = 0, n16lo = 0;      // e16lo: lo halves of an fp16 x 3 stream
};

}  // namespace

constexpr int PIPE_STREAM_DEFAULT = 0;      // (A/B: scripts/stream_ab.sh)
constexpr int RANGE_WORD = 16;      // ctx->dev_status[RANGE_WORD]: fp16 x 3 range flag (a cache line of its own; eonerf_range_status)
constexpr int RANGE_STICKY_WORD = 17;   // ... and the WEIGHT criteria of the last re-pack: reported like RANGE_WORD, cleared only by the next eonerf_set_weights
constexpr int DIGEST_WORD = 32;     // ctx->dev_status[32..35]: two 64-bit ray digests (eonerf_rays.h: ray_word_digest) -- [0] what eonerf_presample's
                                    // sampler read, [1] what the backward of the forward that consumed those samples finds in the same buffers

struct eonerf_ctx {
    eonerf_config cfg;
    int n_samples = 128;  // int(2 / render_step_size) of the next calls: 2 .. 256 (eonerf_set_n_samples; cfg.n_samples at create)
    int prec;             // cfg.precision: EONERF_FP32 / EONERF_BF16 / EONERF_F16X3 (inference only)
    bool bf16;
    int n_cu;
    int enc_pair = 1;       // the shadow pass' encoding products and input-gradient tail in one kernel (eonerf_enc_pair.hip; EONERF_ENC_PAIR=0: ig_tail + two GEMM jobs)
    int pipe_xcd = 0;       // XCD-local pipelines of the pipelined backward (EONERF_PIPE_XCD; BwdPipeArgs::xcd_local)
    int stagger = 0;        // wave stagger of the chain kernels (EONERF_STAGGER; MlpFwdArgs::stagger)
    int wgrad_riders = 1;   // EONERF_WGRAD_RIDERS=0: the sigma row and the embedding columns as jobs of their own (A/B switch)
    int wgrad_items;      // target number of weight-gradient work items per launch (EONERF_WGRAD_ITEMS, default 48 per job)
    ParamLayout pl;
    DevStream fwd_full, fwd_dens, bwd_full, bwd_dens, bwd_rgb, bwd_full_ig, pipe_wt, bwd_full_heads, bwd_rgb_heads, bwd_dens_heads, ig_tail_wt;
    bool pipe = false;               // layer-pipelined trunk backward (bf16 camera pass; EONERF_PIPE=0 switches back to chain + GEMM)
    int n_pipes = 0;
    int stream_blocks = 0;           // EONERF_PIPE_STREAM: workgroups of the pipelined CAMERA launch that run ready weight-gradient GEMM items
    int n_pipes_stream = 0;          // ... and the pipelines that launch keeps: (CUs - stream_blocks) / 7
    float* loss_scratch = nullptr;   // [LOSS_MAX_BLOCKS] per-block partial sums of k_loss + its arrival counter (self-resetting: no memset per step)
    float* fold = nullptr;           // [FOLD_FLOATS (+ 256 x 256: W_bott transposed, for the backward's tail kernel)] fp32: the heads' first layers folded with the bottleneck layer (eonerf_pack.h), re-computed
                                     // by k_fold in front of every re-pack
    bool pipe_fallback = true;       // after a REPORTED watchdog fault the context leaves the pipelined path for good (EONERF_PIPE_FALLBACK=0: stay)
    bool need_repack = false;        // ... and the chain + GEMM path's weight streams have to be packed before the next call
    // Training forwards whose backward is still outstanding: workspace -> the path (pipelined or chain + GEMM) its layout was carved for and
    // its mask slots were written for.  A backward runs in the mode of ITS forward even if eonerf_device_status switched the context in
    // between (the autograd paths keep a workspace across arbitrary host code: render_image chunks, EONerfMLP.rendering)
    std::unordered_map<const void*, bool> ws_pipe;
    int pipe_fault_stage = -1;       // test hook (EONERF_PIPE_FAULT)
    bool deterministic = false;      // EONERF_DETERMINISTIC=1: every atomic flush of the backward is replaced by partials + a fixed-order sum
    bool pipe_partials = false;      // the pipelined launches flush their stationary dW through partial buffers + a reduction kernel (always
                                     // in deterministic mode; EONERF_PIPE_PARTIALS=1 alone: A/B switch against the atomic flush)
    unsigned long long* pipe_stamps = nullptr;   // diagnostics (EONERF_PIPE_STAMPS=1): cycle sums per stage, read by eonerf_debug_pipe_stamps
    uint64_t noise_seed = 0x5eed5eedULL; uint32_t noise_call = 0;   // in-kernel Philox jitter (eonerf_set_noise_seed)
    // eonerf_presample: the camera sampler of the NEXT training forward already ran (under the gradient exchange of the step before);
    // the forward whose arguments and carve match consumes the record, any other forward drops it and samples again
    struct Presample { bool valid = false; const void* ws = nullptr; const float* rays = nullptr; const int64_t* img_idx = nullptr;
                       const float* zsteps = nullptr; const int* count_out = nullptr; int n_rays = 0, flags = 0, n_samples = 0; bool pipe = false; uint32_t call = 0; } pre;
    // two-bucket gradient exchange (eonerf_set_exchange_event): recorded on the backward's stream as soon as the EARLY block of the
    // gradient message (ParamLayout::early) is final; exch_cus CUs are left out of the grids of the gradient kernels launched behind that
    // point, so that the collective's kernel finds a CU while they run (every large kernel here fills the CUs it is given)
    hipEvent_t exch_event = nullptr; int exch_cus = 0; bool exch_recorded = false;
    const void* pre_consumed_ws = nullptr;   // workspace of the training forward that consumed a presample record: its backward checks the ray digest
    bool full_ig_dirty = false;      // packed lazily: only a differentiable EONerfMLP.forward with an input gradient reads it
    int* enc_colmap = nullptr;       // [64] device: encoding slot -> reference column (or -1)
    int* dev_status = nullptr;       // STICKY device status word (watchdog bits of the pipelined backward, bit 8: a remote rank's fault);
                                     // written by the kernels, gates eonerf_adam_step, read and cleared only by eonerf_device_status
    bool weights_set = false;
    bool dens_dirty = false;         // density-only streams are re-packed lazily (only the shadow pass reads them) ...
    bool dens_used = false;          // ... unless the cycle since the last re-pack used them: then they are re-packed with the others (one launch fewer per step)
    // measurement hooks
    int prof_cap = 0;
    std::vector<hipEvent_t> prof_ev[EONERF_PROF_KERNELS][2];
    int prof_n[EONERF_PROF_KERNELS] = {};
};

namespace {

CarveCfg carve_cfg(const eonerf_ctx* ctx) {
    CarveCfg c;
    c.bf16 = ctx->bf16; c.pipe = ctx->pipe; c.deterministic = ctx->deterministic; c.pipe_partials = ctx->pipe_partials;
    c.n_pipes = ctx->n_pipes; c.n_samples = ctx->n_samples;
    c.enc_part_wgs = (ctx->enc_pair && ctx->pipe && !ctx->deterministic) ? ctx->n_cu : 0;
    return c;
}

int upload(DevStream& d, const PackedStream& s) {
    d.bytes = s.bytes; d.n_chunks = (int)s.chunks.size(); d.n16 = (int)s.e16.size(); d.n32 = (int)s.e32.size(); d.n16lo = (int)s.e16lo.size();
    HIP_TRY(hipMalloc(&d.data, s.bytes + 1024));      // + slack: the chain kernels copy whole 1-KiB pieces (WStream::round)
    HIP_TRY(hipMemset(d.data, 0, s.bytes + 1024));
    HIP_TRY(hipMalloc(&d.chunks, s.chunks.size() * sizeof(ChunkDesc)));
    HIP_TRY(hipMemcpy(d.chunks, s.chunks.data(), s.chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    if (d.n16) {
        HIP_TRY(hipMalloc(&d.e16, s.e16.size() * sizeof(PackEntry)));
        HIP_TRY(hipMemcpy(d.e16, s.e16.data(), s.e16.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
    }
    if (d.n32) {
        HIP_TRY(hipMalloc(&d.e32, s.e32.size() * sizeof(PackEntry)));
        HIP_TRY(hipMemcpy(d.e32, s.e32.data(), s.e32.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
    }
    if (d.n16lo) {
        HIP_TRY(hipMalloc(&d.e16lo, s.e16lo.size() * sizeof(PackEntry)));
        HIP_TRY(hipMemcpy(d.e16lo, s.e16lo.data(), s.e16lo.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
    }
    return 0;
}
void release(DevStream& d) {
    if (d.data) (void)hipFree(d.data);
    if (d.chunks) (void)hipFree(d.chunks);
    if (d.e16) (void)hipFree(d.e16);
    if (d.e32) (void)hipFree(d.e32);
    if (d.e16lo) (void)hipFree(d.e16lo);
    d = DevStream();
}

// (re)packs the fp32 master weights into up to PACK_MAX_JOBS packed streams in ONE launch (blockIdx.y = job): after every
// optimizer step three streams x {bf16, fp32} entries are rewritten, and six ~5 us launches cost more than the copies
constexpr int PACK_MAX_JOBS = 16;
struct PackJob { const PackEntry* e; int n; uint8_t* data; int kind; };      // kind: 0 fp32, 1 bf16, 2 fp16 (hi half of a split value), 3 its lo half
struct PackJobs { PackJob j[PACK_MAX_JOBS]; };
// sources >= fold_base come from the fold buffer (ParamLayout::fold_w / fold_b)
__global__ void k_pack(const float* flat, const float* fold, int fold_base, PackJobs jobs, int* range_flag) {
    const PackJob jb = jobs.j[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += gridDim.x * blockDim.x) {
        const PackEntry pe = jb.e[i];
        const float v = pe.src < 0 ? 0.f : (pe.src >= fold_base ? fold[pe.src - fold_base] : flat[pe.src]);
        if (jb.kind == 1) *reinterpret_cast<__bf16*>(jb.data + pe.dst) = (__bf16)v;
        else if (jb.kind == 0) *reinterpret_cast<float*>(jb.data + pe.dst) = v;
        else {
            const _Float16 hi = (_Float16)v;
            *reinterpret_cast<_Float16*>(jb.data + pe.dst) = jb.kind == 2 ? hi : (_Float16)(v - (float)hi);
            if (jb.kind == 2 && !(fabsf(v) <= 65504.f)) atomicOr(range_flag + (RANGE_STICKY_WORD - RANGE_WORD), 1);      // a weight (or folded weight) outside fp16's range, or not finite
        }
    }
}
// (a stream set that needs more than PACK_MAX_JOBS jobs goes out in several launches: today's largest set is 9 streams, 13 jobs)
int pack(const eonerf_ctx* ctx, const std::vector<const DevStream*>& streams, const float* flat, hipStream_t st) {
    PackJobs jobs;
    int n = 0, most = 1;
    auto flush = [&]() {
        if (n) hipLaunchKernelGGL(k_pack, dim3(std::min((most + 255) / 256, 1024), n), dim3(256), 0, st, flat, ctx->fold, (int)ctx->pl.total, jobs, ctx->dev_status + RANGE_WORD);
        n = 0; most = 1;
    };
    for (const DevStream* d : streams) {
        if (n + 3 > PACK_MAX_JOBS) flush();
        if (d->n16) jobs.j[n++] = PackJob{d->e16, d->n16, d->data, d->n16lo ? 2 : 1};
        if (d->n16lo) jobs.j[n++] = PackJob{d->e16lo, d->n16lo, d->data, 3};
        if (d->n32) jobs.j[n++] = PackJob{d->e32, d->n32, d->data, 0};
        most = std::max(most, std::max(d->n16, d->n32));
    }
    flush();
    return (int)hipGetLastError();
}

// The heads' first layers folded with the bottleneck layer (eonerf_pack.h): fold[o][i] = sum_k W_AT[o][k] W_b[k][i], b_f[o] = sum_k
// W_AT[o][k] b_b[k] + b_AT[o], with W_AT = [W_A1; W_T1[:, :256]].  fp32 FMAs in a fixed order (four interleaved partial sums over k:
// deterministic).  Block = 2 output rows, thread = column i: 16.8 M MACs on 128 workgroups in front of every re-pack; the k loop is
// unrolled so that 64 loads of a W_b column are in flight (a batch of 16 was still a chain of L2 round trips: 14 us).
struct FoldArgs { const float *w_a1, *b_a1, *w_t1, *b_t1, *w_b, *b_b; float* fold; float* wbt; };
__global__ __launch_bounds__(256) void k_fold(FoldArgs a) {
    __shared__ float wat[2][256];
    const int o0 = blockIdx.x * 2, i = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int o = o0 + r;
        wat[r][i] = o < 128 ? a.w_a1[(size_t)o * 256 + i] : a.w_t1[(size_t)(o - 128) * 260 + i];
    }
    __syncthreads();
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 16
    for (int k = 0; k < 256; k += 4) {
        float wb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wb[u] = a.w_b[(size_t)(k + u) * 256 + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[0][u] = fmaf(wat[0][k + u], wb[u], acc[0][u]); acc[1][u] = fmaf(wat[1][k + u], wb[u], acc[1][u]); }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) a.fold[(size_t)(o0 + r) * 256 + i] = (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
    // by-product for the backward's tail kernel (bott_wgrad_body): W_bott transposed, two of its rows per block
#pragma unroll
    for (int r = 0; r < 2; ++r) a.wbt[(size_t)i * 256 + o0 + r] = a.w_b[(size_t)(o0 + r) * 256 + i];
    if (i < 2) {
        const int o = o0 + i;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 256; ++k) b4[k & 3] = fmaf(wat[i][k], a.b_b[k], b4[k & 3]);
        a.fold[256 * 256 + o] = ((b4[0] + b4[1]) + (b4[2] + b4[3])) + (o < 128 ? a.b_a1[o] : a.b_t1[o - 128]);
    }
}
// fp16 x 3 contexts, at every re-pack: the weight matrices the split streams are made of must lie where hi + lo fp16 carries them to
// fp32-level accuracy (include/eonerf_hip.h, eonerf_range_status).  One block per matrix: max |w| over the matrix
//   > F16X3_W_MAX (or not finite): an absolute operand error of 2^-25 per activation is amplified beyond what "fp32-level" means
//   < F16X3_W_MIN: the lo halves are fp16 subnormals for the whole matrix, the weights keep < 16 bits relative to the largest one
// -> the range flag is raised and the Python layer renders on an fp32 context instead.
constexpr float F16X3_W_MAX = 64.f, F16X3_W_MIN = 1.f / 512.f;
struct RangeJob { const float* w; int n; int check_min; };
struct RangeJobs { RangeJob j[24]; };
__global__ __launch_bounds__(256) void k_weight_range(RangeJobs jobs, int* range_flag) {
    __shared__ float red[256];
    __shared__ int bad[1];
    const RangeJob jb = jobs.j[blockIdx.x];
    if (threadIdx.x == 0) bad[0] = 0;
    __syncthreads();
    float m = 0.f;
    for (int i = threadIdx.x; i < jb.n; i += 256) {
        const float v = fabsf(jb.w[i]);
        if (!(v <= 3.0e38f)) bad[0] = 1;      // inf / NaN
        else m = fmaxf(m, v);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        const float hi_lim = jb.check_min ? F16X3_W_MAX : 65504.f;       // (operands that are not weights -- the embedding rows -- only have to fit)
        if (bad[0] || mx > hi_lim || (jb.check_min && mx > 0.f && mx < F16X3_W_MIN)) atomicOr(range_flag + (RANGE_STICKY_WORD - RANGE_WORD), 1);
    }
}
int weight_range(const eonerf_ctx* ctx, const float* flat, hipStream_t st) {
    const ParamLayout& pl = ctx->pl;
    RangeJobs jobs;
    int n = 0;
    auto add = [&](int ti, int check_min) { jobs.j[n++] = RangeJob{flat + pl.t[ti].offset, pl.t[ti].rows * pl.t[ti].cols, check_min}; };
    for (int l = 0; l < 8; ++l) add(pl.trunk_w[l], 1);
    add(pl.sig_w, 1); add(pl.a2_w, 1);
    for (int l = 1; l < 4; ++l) add(pl.t_w[l], 1);
    add(pl.tsc_w, 1); add(pl.tbe_w, 1);
    add(pl.t_w[0], 0);      // its embedding columns are packed as they are; its bottleneck columns enter through the fold
    add(pl.emb, 0);
    jobs.j[n++] = RangeJob{ctx->fold, 256 * 256, 1};      // [W_A1; W_T1'] W_bott, what the streams hold of the three folded layers
    hipLaunchKernelGGL(k_weight_range, dim3(n), dim3(256), 0, st, jobs, ctx->dev_status + RANGE_WORD);
    return (int)hipGetLastError();
}

int fold_heads(const eonerf_ctx* ctx, const float* flat, hipStream_t st) {
    const ParamLayout& pl = ctx->pl;
    FoldArgs a{flat + pl.t[pl.a1_w].offset, flat + pl.t[pl.a1_b].offset, flat + pl.t[pl.t_w[0]].offset, flat + pl.t[pl.t_b[0]].offset,
               flat + pl.t[pl.bot_w].offset, flat + pl.t[pl.bot_b].offset, ctx->fold, ctx->fold + FOLD_FLOATS};
    hipLaunchKernelGGL(k_fold, dim3(128), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

// Measurement hook (eonerf_clock_probe): a FIXED amount of dense bf16 MFMA work on every CU -- 4 waves per workgroup (one per SIMD), each
// CLOCK_PROBE_MFMAS v_mfma_f32_32x32x16_bf16 on four independent accumulators -- bracketed, in workgroup 0, by the shader-clock counter
// (s_memtime) and the constant 100-MHz counter (s_memrealtime).  cycles / ticks x 100 = the shader clock in MHz the chip held over the
// probe; the probe's duration (ticks x 10 ns) is the same statement without trusting s_memtime: fixed work, time ~ 1 / clock.
constexpr int CLOCK_PROBE_MFMAS = 8192;
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(16))) float probe_f32x16;
__global__ __launch_bounds__(256) void k_clock_probe(float* out) {
    probe_bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (float)(threadIdx.x + e)); b[e] = (__bf16)(0.002f * (float)(threadIdx.x ^ e)); }
    probe_f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < CLOCK_PROBE_MFMAS / 4; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += (c0[e] + c1[e]) + (c2[e] + c3[e]);
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float cyc = (float)(t1 - t0), ticks = (float)(r1 - r0);
        out[0] = cyc; out[1] = ticks; out[2] = ticks > 0.f ? cyc / ticks * 100.f : 0.f;
        out[3] = s == 12345.678f ? 1.f : 0.f;      // keeps the accumulators live
    }
}

struct ProfScope {      // brackets one kernel launch with events when profiling is on
    eonerf_ctx* c; int k; hipStream_t st; bool on;
    ProfScope(eonerf_ctx* ctx, int kernel, hipStream_t s) : c(ctx), k(kernel), st(s), on(kernel >= 0 && ctx->prof_cap > 0 && ctx->prof_n[kernel] < ctx->prof_cap) {
        if (on) (void)hipEventRecord(c->prof_ev[k][0][c->prof_n[k]], st);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(c->prof_ev[k][1][c->prof_n[k]], st); c->prof_n[k]++; } }
};

inline bool slabs_addressable(const eonerf_ctx* ctx, size_t p_cap) { return slab_blocks_addressable(ctx->bf16, p_cap); }
// a call that writes `ws` ends what eonerf_presample left there
inline void drop_presample(eonerf_ctx* ctx, const void* ws) { if (ctx->pre.valid && ctx->pre.ws == ws) ctx->pre.valid = false; }
// rays per call: n_rays x (n_samples - 1) samples must stay below 2^31
inline bool rays_in_range(const eonerf_ctx* ctx, int n_rays) { return n_rays <= (1 << 24) / (ctx->n_samples > 128 ? ctx->n_samples / 128 : 1); }

CarveCfg carve_cfg(const eonerf_ctx* ctx);
RenderWs carve_render(const eonerf_ctx* ctx, void* base, int n_rays, int flags) { return carve_render(carve_cfg(ctx), base, n_rays, flags); }

void note_train_forward(eonerf_ctx* ctx, const void* ws) {
    if (ctx->ws_pipe.size() > 1024) ctx->ws_pipe.clear();       // forwards that never saw a backward (caller dropped the graph)
    ctx->ws_pipe[ws] = ctx->pipe;
}
struct PipeModeGuard {      // a backward call runs on the path its forward ran on; the context's own choice is restored on return
    eonerf_ctx* c; bool keep;
    PipeModeGuard(eonerf_ctx* ctx, const void* ws) : c(ctx), keep(ctx->pipe), ws_(ws) {
        auto it = c->ws_pipe.find(ws);
        if (it != c->ws_pipe.end()) c->pipe = it->second;
    }
    const void* ws_;
    ~PipeModeGuard() { c->pipe = keep; c->ws_pipe.erase(ws_); }      // the forward's record is spent (a backward runs once per forward)
};

AmbientW ambient_w(const eonerf_ctx* ctx, const float* flat) {
    const ParamLayout& pl = ctx->pl;
    return AmbientW{flat + pl.t[pl.am1_w].offset, flat + pl.t[pl.am1_b].offset, flat + pl.t[pl.am2_w].offset, flat + pl.t[pl.am2_b].offset};
}

int ensure_density_streams(eonerf_ctx* ctx, const float* flat, hipStream_t st) {
    ctx->dens_used = true;
    if (!ctx->dens_dirty) return 0;
    const int rc = ctx->prec == EONERF_F16X3 ? pack(ctx, {&ctx->fwd_dens}, flat, st)
                 : ctx->pipe ? pack(ctx, {&ctx->fwd_dens, &ctx->bwd_dens, &ctx->bwd_dens_heads, &ctx->ig_tail_wt}, flat, st)
                             : pack(ctx, {&ctx->fwd_dens, &ctx->bwd_dens}, flat, st);
    if (!rc) ctx->dens_dirty = false;
    return rc;
}

int run_mlp_fwd(eonerf_ctx* ctx, const PassBuffers& b, const float* flat, int p_cap, bool full, int mode, hipStream_t st, int prof_id = -1, bool render_train = false) {
    if (!full) { const int rc = ensure_density_streams(ctx, flat, st); if (rc) return rc; }
    const DevStream& ds = full ? ctx->fwd_full : ctx->fwd_dens;
    MlpFwdArgs a;
    a.px = b.px; a.py = b.py; a.pz = b.pz; a.simg = b.simg;
    a.emb = flat + ctx->pl.t[ctx->pl.emb].offset;
    a.n_pts = b.n_pts; a.p_pad = p_cap;
    a.stream = ds.data; a.chunks = ds.chunks; a.n_chunks = ds.n_chunks;
    a.sigma = b.sigma; a.albedo = b.albedo; a.ts = b.ts; a.tb = b.tb;
    a.act = b.act; a.masks = b.masks;
    a.range_flag = ctx->dev_status + RANGE_WORD;
    a.stagger = ctx->stagger;
    // training passes of the render path with the pipelined backward: the trunk's ReLU' comes from the X images, only the heads chain
    // reads mask bits (slot 7 = X_8 for its last layer)
    a.mask_from = (render_train && ctx->pipe) ? 7 : 0;
    const int tile = ctx->bf16 ? PBf16::TILE : PF32::TILE;
    const int grid = std::min(ctx->n_cu, p_cap / tile);
    if (ctx->prec == EONERF_F16X3 && mode != 0) return EONERF_E_UNSUPPORTED;      // the split precision is an inference precision
    if (prof_id < 0) return (int)eo_launch_mlp_fwd(a, ctx->prec, full, mode, grid, st);
    ProfScope ps(ctx, prof_id, st);
    return (int)eo_launch_mlp_fwd(a, ctx->prec, full, mode, grid, st);
}


// One pipelined launch of a backward call (eonerf_bwd_pipe.hip).  `slot` = which of the PIPE_LAUNCHES sync blocks it uses; the FIRST
// pipelined launch of the call (first_of_backward) zeroes all of them together with the GEMM's accumulator and work queue, which sit
// right in front (one memset for everything the backward needs zeroed).  A watchdog that fires goes to the context's STICKY status
// word (ctx->dev_status), which no launch clears.
void pipe_common(eonerf_ctx* ctx, const RenderWs& w, const PassBuffers& b, int p_cap, float* d_flat, int slot, int n_pipes, int n_stages, BwdPipeArgs& pa) {
    memset(&pa, 0, sizeof(pa));
    uint32_t* sync = w.pipe.sync + (size_t)slot * (w.pipe.sync_bytes / sizeof(uint32_t));
    pa.n_pts = b.n_pts; pa.p_pad = p_cap; pa.n_pipes = n_pipes; pa.n_stages = n_stages;
    pa.act = b.act; pa.grd = b.grd;
    pa.rings = w.pipe.rings; pa.role_counter = reinterpret_cast<int*>(sync) + 32; pa.error = ctx->dev_status;
    pa.scratch_word = sync + 64; pa.flags = sync + 64 + (size_t)n_pipes * n_stages * 32;
    pa.d_flat = d_flat; pa.fault_stage = -1; pa.stamps = nullptr;
    pa.partials = w.det.pipe_part;
}
int pipe_clear(const RenderWs& w, bool first_of_backward, int slot, hipStream_t st) {
    if (first_of_backward) {
        uint8_t* lo = reinterpret_cast<uint8_t*>(w.m_bott);
        uint8_t* hi = reinterpret_cast<uint8_t*>(w.pipe.sync) + PIPE_LAUNCHES * w.pipe.sync_bytes;
        HIP_TRY(hipMemsetAsync(lo, 0, (size_t)(hi - lo), st));
    }
    (void)slot;
    return 0;
}

// CUs the pipelined launch leaves without a stage (256 - 7 x 36 = 4): enough of them, and the per-ray ambient-head backward rides there
inline int pipe_spare_cus(const eonerf_ctx* ctx) {
    const int spare = ctx->n_cu - ctx->n_pipes * PIPE_STAGES;
    return (ctx->pipe && !ctx->deterministic && spare >= 2) ? std::min(spare, 8) : 0;
}

// trunk layers 7..1 of one pass: dX chain + weight gradients.  Reads dY_7 from w.pipe.dy_in (written by the heads chain or the heads
// pipeline), accumulates dW / db of the trunk into d_flat, saves dY_5 / dY_0 in b.grd.
struct WgradPlan;
int fill_stream_args(const eonerf_ctx* ctx, const WgradPlan& plan, int* queue, int* stop, PipeStreamArgs& sa);
int run_bwd_pipe(eonerf_ctx* ctx, const RenderWs& w, const PassBuffers& b, int p_cap, float* d_flat, int prof_id, hipStream_t st, int slot, bool first_of_backward,
                 const AmbientBwdArgs* amb = nullptr, const WgradPlan* plan = nullptr) {
    const ParamLayout& pl = ctx->pl;
    { const int rc = pipe_clear(w, first_of_backward, slot, st); if (rc) return rc; }
    ProfScope ps(ctx, prof_id, st);
    BwdPipeArgs pa;
    pipe_common(ctx, w, b, p_cap, d_flat, slot, plan ? ctx->n_pipes_stream : ctx->n_pipes, PIPE_STAGES, pa);
    pa.wt = ctx->pipe_wt.data; pa.dy_in = w.pipe.dy_in;
    pa.xcd_local = ctx->pipe_xcd;
    pa.fault_stage = ctx->pipe_fault_stage; pa.stamps = ctx->pipe_stamps;
    if (amb) { pa.amb = *amb; pa.amb_blocks = pipe_spare_cus(ctx); }
    for (int s = 0; s < PIPE_STAGES; ++s) {
        const int l = 7 - s;
        pa.dw_off[s] = pl.t[pl.trunk_w[l]].offset; pa.db_off[s] = pl.t[pl.trunk_b[l]].offset; pa.dw_ld[s] = l == 5 ? 319 : 256;
    }
    if (plan) {      // streaming roles: ready items of the weight-gradient GEMM under the stages
        PipeStreamArgs sa;
        uint32_t* sync = w.pipe.sync + (size_t)slot * (w.pipe.sync_bytes / sizeof(uint32_t));
        const int rcs = fill_stream_args(ctx, *plan, w.queue, reinterpret_cast<int*>(sync) + 48, sa);      // (word 48 of the launch's zeroed sync header)
        if (rcs) return rcs;
        HIP_TRY(eo_launch_bwd_pipe_stream(pa, sa, st));
        return 0;
    }
#ifdef EO_COR
    // Co-residency falsifier (diagnostic builds only, scripts/coresidency.sh): EONERF_COR_PARTNER=1 launches the dummy streaming partner
    // (eonerf_bwd_pipe.hip: k_cor_partner) on a side stream BESIDE the camera pass' pipelined launch, =2 on the same stream in front of it
    // (the partner alone on the chip); EONERF_COR_GB = bytes it streams (default 2 GB, out of the camera pass' activation slab).
    // Its durations are summed and printed when the process ends.
    {
        static const int mode = getenv("EONERF_COR_PARTNER") ? atoi(getenv("EONERF_COR_PARTNER")) : 0;
        static const double gb = getenv("EONERF_COR_GB") ? atof(getenv("EONERF_COR_GB")) : 2.0;
        struct Cor { hipStream_t side = nullptr; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double gb = 0; int mode = 0;
                     ~Cor() { double ms = 0; int n = 0; for (auto& e : ev) { float t = 0; if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) { ms += t; ++n; } }
                              if (n) fprintf(stderr, "[cor] partner mode %d: %d launches, avg %.4f ms, %.1f GB/s (%.2f GB each)\n", mode, n, ms / n, gb / (ms / n * 1e-3), gb); } };
        static Cor cor;
        if (mode && prof_id == EONERF_PROF_BWD_PIPE_CAMERA) {
            cor.gb = gb; cor.mode = mode;
            if (!cor.side) HIP_TRY(hipStreamCreateWithFlags(&cor.side, hipStreamNonBlocking));
            hipEvent_t e0, e1, fork;
            HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
            hipStream_t ps = mode == 1 ? cor.side : st;
            if (mode == 1) { HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); HIP_TRY(hipEventRecord(fork, st)); HIP_TRY(hipStreamWaitEvent(cor.side, fork, 0)); }
            const size_t slab_bytes = (size_t)p_cap * 2048 * 2;      // (X_1 .. X_8 rows of the camera pass' activation slab: >= 2 GB at 4096 rays)
            HIP_TRY(hipEventRecord(e0, ps));
            HIP_TRY(eo_launch_cor_partner(b.act, slab_bytes, (size_t)(gb * 1e9), ctx->n_cu, ps));
            HIP_TRY(hipEventRecord(e1, ps));
            if (cor.ev.size() < 4096) cor.ev.push_back({e0, e1});
        }
    }
#endif
    HIP_TRY(eo_launch_bwd_pipe(pa, st));
    if (pa.partials) HIP_TRY(eo_launch_pipe_reduce(pa, st));
    return 0;
}

// Weight gradients of up to two MLP passes in ONE split-K launch (eonerf_wgrad.hip) + the bottleneck factor product:
//   full: a pass through the whole field (camera pass / EONerfMLP.forward), with or without the transient head in the graph;
//   dens: a density-only pass (shadow pass / query_density).  Either may be null.  Gradients are ACCUMULATED into d_flat.
// plan != nullptr: the table is only BUILT (into plan->tab; the camera pass' layer-0 / skip-column jobs, whose dY_0 / dY_5 operands the
// pipelined camera launch is still to write, at the END of the queue: plan->ready_items = the items in front of them) -- the caller
// launches the pipelined trunk with streaming roles on the ready items and then launch_planned_wgrad() for what is left.
struct WgradPlan { WgradJobTable tab; int ready_items; };
int fill_stream_args(const eonerf_ctx* ctx, const WgradPlan& plan, int* queue, int* stop, PipeStreamArgs& sa) {
    if (plan.tab.n > WGRAD_STREAM_JOBS) return EONERF_E_STATE;
    memset(&sa, 0, sizeof(sa));
    sa.blocks = ctx->stream_blocks; sa.ready_items = plan.ready_items; sa.queue = queue; sa.stop = stop;
    for (int i = 0; i < plan.tab.n; ++i) sa.tab.j[i] = plan.tab.j[i];
    sa.tab.aux = plan.tab.aux; sa.tab.n = plan.tab.n; sa.tab.items = plan.tab.items;
    return 0;
}
int launch_planned_wgrad(eonerf_ctx* ctx, const WgradPlan& plan, int p_cap, int* queue, hipStream_t st) {
    ProfScope ps(ctx, EONERF_PROF_WGRAD, st);
    HIP_TRY(eo_launch_wgrad(plan.tab, ctx->n_cu - (ctx->exch_event ? ctx->exch_cus : 0), p_cap, queue, ctx->bf16, st, nullptr, false));
    return 0;
}
int run_weight_gradients(eonerf_ctx* ctx, const float* flat, float* d_flat, const PassBuffers* full, bool transient,
                         const PassBuffers* dens, int p_cap, float* m_bott, int* queue, hipStream_t st, bool full_trunk_done = false,
                         bool dens_trunk_done = false, float* det_partials = nullptr, bool zeroed = false, BottWgradArgs* defer_bott = nullptr,
                         WgradPlan* plan = nullptr, bool dens_enc_done = false) {
    // defer_bott != nullptr: the products that follow from the bottleneck factors are NOT launched here; their arguments are handed back
    // (the render path runs them in one launch with the embedding and ambient-head gradients, eo_launch_step_tail)
    const ParamLayout& pl = ctx->pl;
    auto dptr = [&](int ti) { return d_flat + pl.t[ti].offset; };
    WgradJobTable tab;
    bool late[WGRAD_MAX_JOBS] = {};      // (plan) jobs whose operands the pipelined camera launch writes
    tab.n = 0;
    memset(&tab.aux, 0, sizeof(tab.aux));
    tab.aux.job = -1;
    // riders of the bottleneck-factor job (WgradAux): the sigma row and the embedding columns of the camera pass travel with the job that
    // streams X_8 / dY_T1 anyway.  Not in deterministic mode (its partial-sum tiles have no room for them)
    const bool riders = full && !det_partials && ctx->wgrad_riders;
    const size_t n_tiles = (size_t)p_cap / (ctx->bf16 ? 32 : 16);      // sample tiles of the slabs (block-major layout, eonerf_common.h)
    auto seg0 = [&](const void* slab, SlabBlk blk, int row) {           // (row, sample tile 0)
        return reinterpret_cast<const uint8_t*>(slab) + ((size_t)blk.s * n_tiles + (row - blk.s)) * SEG_B;
    };
    auto add = [&](const PassBuffers& b, int grd_row, int m_rows, int act_row, int n_rows, float* dw, int dw_ld, float* db,
                   const int* cmap, int gm, int gn, int wm, int wn) {
        WgradJob& j = tab.j[tab.n++];
        const SlabBlk ba = GrdMap::block(grd_row), bb = ActMap::block(act_row);
        j.a = seg0(b.grd, ba, grd_row); j.b = seg0(b.act, bb, act_row); j.dw = dw; j.db = db; j.col_map = cmap; j.n_pts = b.n_pts;
        j.a_stride = (uint32_t)(ba.r * SEG_B);       // consecutive sample tiles of a block are contiguous
        j.b_stride = (uint32_t)(bb.r * SEG_B);
        j.m_rows = m_rows; j.n_rows = n_rows; j.dw_ld = dw_ld; j.gm = gm; j.gn = gn; j.wm = wm; j.wn = wn;
        j.dw2 = nullptr; j.db2 = nullptr; j.split = m_rows; j.dw2_ld = 0; j.a_units = 0;
    };
    auto split_at = [&](int row, float* dw2, int dw2_ld, float* db2) {      // rows >= row of the job just added go to a second layer's gradient
        WgradJob& j = tab.j[tab.n - 1];
        j.split = row; j.dw2 = dw2; j.dw2_ld = dw2_ld; j.db2 = db2;
    };
    // pipelined: the 256 x 256 products of layers 1..7 (and their biases) were accumulated by the layer-pipelined trunk backward;
    // what is left are the two 256 x 64 products against the encoding (layer 0, skip columns of layer 5) and the sigma row
    auto trunk_jobs = [&](const PassBuffers& b, bool pipelined, bool sigma_job, bool written_late, bool enc_jobs) {
        // (pipelined: dY_0 and dY_5 lie in their slab tiles in unit order -- written once by the stages of layers 1 and 6)
        // (enc_jobs = false: the two products against the encoding were formed by eo_launch_enc_pair, with the pass' input-gradient tail)
        if (enc_jobs) {
        add(b, GRD_ROW_Y0, 256, ACT_ROW_ENC, 64, dptr(pl.trunk_w[0]), 63, dptr(pl.trunk_b[0]), ctx->enc_colmap, 4, 2, 2, 1);
        tab.j[tab.n - 1].a_units = pipelined; late[tab.n - 1] = written_late;
        }
        for (int l = 1; l < 8; ++l) {
            const int in_ld = l == 5 ? 319 : 256;
            if (!pipelined)
                add(b, GRD_ROW_Y0 + 256 * l, 256, ACT_ROW_X1 + 256 * (l - 1), 256, dptr(pl.trunk_w[l]), in_ld, dptr(pl.trunk_b[l]), nullptr, 2, 4, 4, 2);
            if (l == 5 && enc_jobs)   // skip columns 256..318 <- encoding slots
                { add(b, GRD_ROW_Y0 + 256 * 5, 256, ACT_ROW_ENC, 64, dptr(pl.trunk_w[5]) + 256, 319, nullptr, ctx->enc_colmap, 4, 2, 2, 1); tab.j[tab.n - 1].a_units = pipelined; late[tab.n - 1] = written_late; }
        }
        if (sigma_job) add(b, GRD_ROW_SIG, 1, ACT_ROW_X1 + 256 * 7, 256, dptr(pl.sig_w), 256, dptr(pl.sig_b), nullptr, 1, 8, 1, 1);
    };
    if (full) {
        const PassBuffers& c = *full;
        trunk_jobs(c, full_trunk_done, !riders, plan != nullptr, true);
        // bottleneck factors M_a = dA1^T X8 (and M_t = dT1^T X8) + the bias gradients db_A1 (db_T1), finished by eo_launch_bott_wgrad
        // below into THREE weight gradients: the bottleneck layer's and the two head layers' that read the bottleneck output (which is
        // therefore never saved by the forward, nor read back here: see BottWgradArgs)
        if (!zeroed) HIP_TRY(hipMemsetAsync(m_bott, 0, BOTT_SCRATCH_F * sizeof(float), st));
        float* db_at = m_bott + 2 * 128 * 256;
        // dY A1 and dY T1 are the two halves of one 256-row block of the gradient slab: with the transient head both factors are ONE job
        if (transient) {
            add(c, GRD_ROW_A1, 256, ACT_ROW_X1 + 256 * 7, 256, m_bott, 256, db_at, nullptr, 2, 4, 4, 2);          // [M_a; M_t], [db_A1; db_T1]
        } else {
            add(c, GRD_ROW_A1, 128, ACT_ROW_X1 + 256 * 7, 256, m_bott, 256, db_at, nullptr, 2, 4, 2, 2);
        }
        if (riders) {
            WgradAux& x = tab.aux;
            x.job = tab.n - 1;
            x.a2 = seg0(c.grd, GrdMap::block(GRD_ROW_SIG), GRD_ROW_SIG); x.a2_stride = (uint32_t)(GrdMap::block(GRD_ROW_SIG).r * SEG_B);
            x.dw_sig = dptr(pl.sig_w); x.db_sig = dptr(pl.sig_b);
            if (transient) {
                x.b2 = seg0(c.act, ActMap::block(ACT_ROW_EMB), ACT_ROW_EMB); x.b2_stride = (uint32_t)(ActMap::block(ACT_ROW_EMB).r * SEG_B);
                x.dw_emb = dptr(pl.t_w[0]) + 256; x.emb_ld = 260; x.emb_row0 = 128;
            }
        }
        add(c, GRD_ROW_A2, 3, ACT_ROW_A1, 128, dptr(pl.a2_w), 128, dptr(pl.a2_b), nullptr, 1, 4, 1, 1);
        if (transient) {
            if (!riders) add(c, GRD_ROW_T1, 128, ACT_ROW_EMB, 4, dptr(pl.t_w[0]) + 256, 260, nullptr, nullptr, 4, 1, 1, 1);
            for (int l = 1; l < 4; ++l)
                add(c, GRD_ROW_T1 + 128 * l, 128, ACT_ROW_T1 + 128 * (l - 1), 128, dptr(pl.t_w[l]), 128, dptr(pl.t_b[l]), nullptr, 2, 4, 2, 1);
            add(c, GRD_ROW_T5, 2, ACT_ROW_T1 + 384, 128, dptr(pl.tsc_w), 128, dptr(pl.tsc_b), nullptr, 1, 4, 1, 1);     // row 0: d ts_pre, row 1: d tb_pre
            split_at(1, dptr(pl.tbe_w), 128, dptr(pl.tbe_b));
        }
    }
    if (dens) trunk_jobs(*dens, dens_trunk_done, true, false, !dens_enc_done);
    if (plan) {      // late jobs to the end of the table (stable), the riders' job index follows
        WgradJob ordered[WGRAD_MAX_JOBS];
        int k = 0, aux_job = tab.aux.job;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < tab.n; ++i)
                if (late[i] == (pass == 1)) { if (i == tab.aux.job) aux_job = k; ordered[k++] = tab.j[i]; }
        for (int i = 0; i < tab.n; ++i) tab.j[i] = ordered[i];
        tab.aux.job = aux_job;
    }
    // every work item = one slice of one job's sample range, equal slices for every job; persistent workgroups pull items from one
    // counter.  Since the K loop is instantiated per tile shape the launch is HBM-bound (5.7 TB/s) and an item's time follows the
    // bytes its job moves per K step.  Measured rules (scripts/wgrad_items_sweep.sh; slices in proportion to the bytes were tried
    // twice and lost to equal slices at every item count):
    //   few jobs (the rgb state: 4): the HEAVY jobs' items (>= 45 % of the heaviest job's bytes per step) fill exactly one round,
    //     just under one item per CU, and the light jobs' items fill the gaps behind them: 0.227-0.238 ms at 288-320 items against
    //     0.26 at 512 and 0.28 at 352 (heavy items spill into a second round)
    //   many jobs (the full state with the pipelined trunk: 11): ~2.5 items per CU: flat (0.472-0.478 ms) from 650 to 830 items,
    //     0.49-0.50 at 512-600 and at 1,024
#ifdef EO_WGRAD_MASK
    {   // DIAGNOSTIC BUILDS ONLY (scripts/wgrad_jobs.sh builds with -DEO_WGRAD_MASK): EONERF_WGRAD_MASK keeps only the jobs whose bit is set --
        // gradients are WRONG, timing only.  The shipped library does not read the variable.
        static const char* mk = getenv("EONERF_WGRAD_MASK");
        if (mk) {
            const unsigned long mask = strtoul(mk, nullptr, 0);
            int k = 0;
            for (int i = 0; i < tab.n; ++i) if (mask & (1ul << i)) { if (i == tab.aux.job) tab.aux.job = k; else if (k == tab.aux.job && i != k) tab.aux.job = -1; tab.j[k] = tab.j[i]; late[k] = late[i]; ++k; }
            if (tab.aux.job >= k) tab.aux.job = -1;
            tab.n = k;
        }
    }
#endif
    double wmax = 0.0, wj[WGRAD_MAX_JOBS];
    for (int k = 0; k < tab.n; ++k) {
        wj[k] = (double)((tab.j[k].m_rows + 15) / 16 * 16 + (tab.j[k].n_rows + 15) / 16 * 16) * SEG_B;
        wmax = std::max(wmax, wj[k]);
    }
    int n_heavy = 0;
    for (int k = 0; k < tab.n; ++k) n_heavy += wj[k] >= 0.45 * wmax ? 1 : 0;
    tab.items = 0;
    for (int k = 0; k < tab.n; ++k) {
        WgradJob& j = tab.j[k];
        int fill = (int)(2.54 * ctx->n_cu / tab.n + 0.5);
        if (tab.n <= 8) fill = (ctx->n_cu - 1) / std::max(n_heavy, 1);
        int sl = ctx->wgrad_items ? (ctx->wgrad_items + tab.n / 2) / tab.n : (tab.n > 16 ? std::max(fill, 48) : std::min(std::max(fill, 1), 256));
        if (det_partials && sl > 48) sl = 48;       // the partial buffer holds WGRAD_MAX_JOBS x 48 items
        j.slices = sl < 1 ? 1 : sl;
    }
    for (int k = 0; k < tab.n; ++k) { tab.j[k].item0 = tab.items; tab.items += tab.j[k].slices; }
    if (plan) {
        int n_late = 0;
        for (int i = 0; i < tab.n; ++i) n_late += late[i] ? 1 : 0;
        plan->tab = tab;
        plan->ready_items = tab.n > n_late ? tab.j[tab.n - n_late].item0 : 0;
        if (n_late == 0) plan->ready_items = tab.items;
    } else {
        ProfScope ps(ctx, EONERF_PROF_WGRAD, st);
        HIP_TRY(eo_launch_wgrad(tab, ctx->n_cu - (ctx->exch_event ? ctx->exch_cus : 0), p_cap, queue, ctx->bf16, st, det_partials, !zeroed));
    }
    if (full) {   // the three weight gradients that follow from the bottleneck factors the GEMM above accumulated
        BottWgradArgs bw;
        bw.w_a1 = flat + pl.t[pl.a1_w].offset; bw.m_a = m_bott; bw.db_at = m_bott + 2 * 128 * 256;
        bw.w_t1 = transient ? flat + pl.t[pl.t_w[0]].offset : nullptr; bw.m_t = m_bott + 128 * 256;
        bw.w_bott_t = ctx->fold + FOLD_FLOATS; bw.b_bott = flat + pl.t[pl.bot_b].offset;       // (the transposed copy is as current as the packed streams)
        bw.d_w = dptr(pl.bot_w); bw.d_b = dptr(pl.bot_b);
        bw.d_w_a1 = dptr(pl.a1_w); bw.d_b_a1 = dptr(pl.a1_b);
        bw.d_w_t1 = transient ? dptr(pl.t_w[0]) : nullptr; bw.d_b_t1 = transient ? dptr(pl.t_b[0]) : nullptr;
        if (defer_bott) *defer_bott = bw;
        else HIP_TRY(eo_launch_bott_wgrad(bw, st));
    }
    return 0;
}

}  // namespace

extern "C" {

int eonerf_version(void) { return EONERF_VERSION; }

const char* eonerf_strerror(int code) {
    switch (code) {
        case EONERF_OK: return "ok";
        case EONERF_E_ARG: return "eonerf: invalid argument";
        case EONERF_E_WORKSPACE: return "eonerf: workspace too small";
        case EONERF_E_STATE: return "eonerf: call sequence error (set_weights / train forward missing, or a ray buffer refilled between eonerf_presample and its forward)";
        case EONERF_E_UNSUPPORTED: return "eonerf: unsupported configuration";
        case EONERF_E_RANGE: return "eonerf: a weight, an activation or a position left fp16's range (|v| > 65504 or not finite) in an fp16 x 3 call since the last check; its outputs are invalid -- render in fp32";
        case EONERF_E_DEVICE: return "eonerf: a device-side hand-off timed out (pipelined backward watchdog) on this or another rank; the gradients of that step are invalid and every optimizer update since has been skipped";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "eonerf: unknown error";
    }
}

int eonerf_create(eonerf_ctx** out, const eonerf_config* cfg) {
    if (!out || !cfg || cfg->n_images < 1) return EONERF_E_ARG;
    if (cfg->n_samples < 2 || cfg->n_samples > 256) return EONERF_E_UNSUPPORTED;      // (a ray's samples live in the 64 lanes x 4 slots of one wavefront)
    if (cfg->precision != EONERF_FP32 && cfg->precision != EONERF_BF16 && cfg->precision != EONERF_F16X3) return EONERF_E_ARG;
    eonerf_ctx* ctx = new (std::nothrow) eonerf_ctx();
    if (!ctx) return EONERF_E_ARG;
    ctx->cfg = *cfg;
    ctx->n_samples = cfg->n_samples;
    ctx->prec = cfg->precision;
    ctx->bf16 = cfg->precision == EONERF_BF16;
    const bool infer_only = cfg->precision == EONERF_F16X3;      // forward streams only
    { const char* e = getenv("EONERF_DETERMINISTIC"); ctx->deterministic = e && atoi(e) != 0; }
    { const char* e = getenv("EONERF_PIPE_PARTIALS"); ctx->pipe_partials = ctx->deterministic || (e && atoi(e) != 0); }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete ctx; return (int)hipErrorNoDevice; }
    ctx->n_cu = prop.multiProcessorCount;
    { const char* e = getenv("EONERF_WGRAD_ITEMS"); ctx->wgrad_items = e && atoi(e) > 0 ? atoi(e) : 0; }
    { const char* e = getenv("EONERF_WGRAD_RIDERS"); if (e) ctx->wgrad_riders = atoi(e); }
    { const char* e = getenv("EONERF_STAGGER"); if (e) ctx->stagger = atoi(e); }
    { const char* e = getenv("EONERF_PIPE_XCD"); if (e) ctx->pipe_xcd = atoi(e); }
    { const char* e = getenv("EONERF_ENC_PAIR"); if (e) ctx->enc_pair = atoi(e); }
    ctx->pl.build(cfg->n_images);
    int rc = upload(ctx->fwd_full, build_fwd_stream(ctx->pl, ctx->prec, true));
    if (!rc) rc = upload(ctx->fwd_dens, build_fwd_stream(ctx->pl, ctx->prec, false));
    if (!rc && !infer_only) rc = upload(ctx->bwd_full, build_bwd_stream(ctx->pl, ctx->bf16, true, false));
    if (!rc && !infer_only) rc = upload(ctx->bwd_dens, build_bwd_stream(ctx->pl, ctx->bf16, false, true));
    if (!rc && !infer_only) rc = upload(ctx->bwd_rgb, build_bwd_stream(ctx->pl, ctx->bf16, true, false, false));
    if (!rc && !infer_only) rc = upload(ctx->bwd_full_ig, build_bwd_stream(ctx->pl, ctx->bf16, true, true, true));
    {
        const char* e = getenv("EONERF_PIPE");
        ctx->n_pipes = ctx->n_cu / PIPE_STAGES;
        ctx->pipe = ctx->bf16 && ctx->n_pipes >= 1 && !(e && atoi(e) == 0);
        // Residency: the pipelined launch needs its 7 x n_pipes workgroups on the chip AT THE SAME TIME (one per CU: 128 KB of LDS, the
        // whole register file).  Where that cannot hold by construction -- the kernel does not fit a CU of this device, or the process
        // was given a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK: the runtime still reports every CU) -- the chain + GEMM path is used
        // from the start instead of timing out in the first step.  A co-tenant on the card cannot be seen from here: see eonerf_device_status.
        if (ctx->pipe && !(e && atoi(e) == 1)) {
            if (!eo_bwd_pipe_fits_a_cu() || getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) ctx->pipe = false;
        }
        { const char* fb = getenv("EONERF_PIPE_FALLBACK"); ctx->pipe_fallback = !(fb && atoi(fb) == 0); }
        {
            const char* sb = getenv("EONERF_PIPE_STREAM");
            ctx->stream_blocks = sb ? atoi(sb) : PIPE_STREAM_DEFAULT;
            if (ctx->stream_blocks < 0 || ctx->stream_blocks > ctx->n_cu - PIPE_STAGES || ctx->deterministic || ctx->pipe_partials) ctx->stream_blocks = 0;
            ctx->n_pipes_stream = std::min(ctx->n_pipes, (ctx->n_cu - ctx->stream_blocks) / PIPE_STAGES);
        }
        if (!rc && ctx->pipe) rc = upload(ctx->pipe_wt, build_pipe_stream(ctx->pl));
        if (!rc && ctx->pipe) rc = upload(ctx->bwd_full_heads, build_bwd_stream(ctx->pl, true, true, false, true, 1));
        if (!rc && ctx->pipe) rc = upload(ctx->bwd_rgb_heads, build_bwd_stream(ctx->pl, true, true, false, false, 1));
        if (!rc && ctx->pipe) rc = upload(ctx->bwd_dens_heads, build_bwd_stream(ctx->pl, true, false, true, false, 1));
        if (!rc && ctx->pipe) rc = upload(ctx->ig_tail_wt, build_ig_tail_stream(ctx->pl));
        { const char* f = getenv("EONERF_PIPE_FAULT"); ctx->pipe_fault_stage = f ? atoi(f) : -1; }
        { const char* f = getenv("EONERF_PIPE_STAMPS");
          if (!rc && ctx->pipe && f && atoi(f)) rc = (int)hipMalloc(&ctx->pipe_stamps, (size_t)ctx->n_pipes * PIPE_STAGES * 128 * sizeof(unsigned long long)); }
    }
    if (!rc) rc = (int)hipMalloc(&ctx->fold, (FOLD_FLOATS + 256 * 256) * sizeof(float));      // + W_bott transposed (k_fold)
    if (!rc) rc = (int)hipMalloc(&ctx->loss_scratch, (LOSS_MAX_BLOCKS + 4) * sizeof(float));
    if (!rc) rc = (int)hipMemset(ctx->loss_scratch, 0, (LOSS_MAX_BLOCKS + 4) * sizeof(float));
    if (!rc) rc = (int)hipMalloc(&ctx->dev_status, 64 * sizeof(int));
    if (!rc) rc = (int)hipMemset(ctx->dev_status, 0, 64 * sizeof(int));
    if (!rc) {
        int cm[64];
        for (int s = 0; s < 64; ++s) cm[s] = enc_col_of_slot(ctx->bf16, s);
        rc = (int)hipMalloc(&ctx->enc_colmap, sizeof(cm));
        if (!rc) rc = (int)hipMemcpy(ctx->enc_colmap, cm, sizeof(cm), hipMemcpyHostToDevice);
    }
    if (rc) { eonerf_destroy(ctx); return rc; }
    *out = ctx;
    return EONERF_OK;
}

int eonerf_profile_enable(eonerf_ctx* ctx, int max_launches) {
    if (!ctx || max_launches < 0) return EONERF_E_ARG;
    for (int k = 0; k < EONERF_PROF_KERNELS; ++k) {
        for (int s = 0; s < 2; ++s) {
            for (hipEvent_t e : ctx->prof_ev[k][s]) (void)hipEventDestroy(e);
            ctx->prof_ev[k][s].clear();
            for (int i = 0; i < max_launches; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ctx->prof_ev[k][s].push_back(e); }
        }
        ctx->prof_n[k] = 0;
    }
    ctx->prof_cap = max_launches;
    return EONERF_OK;
}

int eonerf_clock_probe(eonerf_ctx* ctx, float* out4, void* stream) {
    if (!ctx || !out4) return EONERF_E_ARG;
    hipLaunchKernelGGL(k_clock_probe, dim3(ctx->n_cu), dim3(256), 0, (hipStream_t)stream, out4);
    return (int)hipGetLastError();
}

const char* eonerf_profile_name(int kernel) {
    static const char* const names[EONERF_PROF_KERNELS] = {"fwd_chain_camera", "bwd_chain_camera", "wgrad_gemm", "fwd_chain_sun", "bwd_chain_sun",
                                                           "bwd_pipe_camera", "bwd_pipe_sun", "ig_tail_sun"};
    return kernel >= 0 && kernel < EONERF_PROF_KERNELS ? names[kernel] : nullptr;
}

int eonerf_profile_read(eonerf_ctx* ctx, int kernel, float* total_ms, int* launches) {
    if (!ctx || kernel < 0 || kernel >= EONERF_PROF_KERNELS || !total_ms || !launches) return EONERF_E_ARG;
    float sum = 0.f;
    for (int i = 0; i < ctx->prof_n[kernel]; ++i) {
        HIP_TRY(hipEventSynchronize(ctx->prof_ev[kernel][1][i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[kernel][0][i], ctx->prof_ev[kernel][1][i]));
        sum += ms;
    }
    *total_ms = sum; *launches = ctx->prof_n[kernel];
    return EONERF_OK;
}

int eonerf_destroy(eonerf_ctx* ctx) {
    if (!ctx) return EONERF_E_ARG;
    for (int k = 0; k < EONERF_PROF_KERNELS; ++k) for (int s = 0; s < 2; ++s) for (hipEvent_t e : ctx->prof_ev[k][s]) (void)hipEventDestroy(e);
    release(ctx->fwd_full); release(ctx->fwd_dens); release(ctx->bwd_full); release(ctx->bwd_dens); release(ctx->bwd_rgb); release(ctx->bwd_full_ig); release(ctx->pipe_wt); release(ctx->bwd_full_heads); release(ctx->bwd_rgb_heads); release(ctx->bwd_dens_heads); release(ctx->ig_tail_wt);
    if (ctx->fold) (void)hipFree(ctx->fold);
    if (ctx->loss_scratch) (void)hipFree(ctx->loss_scratch);
    if (ctx->enc_colmap) (void)hipFree(ctx->enc_colmap);
    if (ctx->pipe_stamps) (void)hipFree(ctx->pipe_stamps);
    if (ctx->dev_status) (void)hipFree(ctx->dev_status);
    delete ctx;
    return EONERF_OK;
}

int eonerf_param_tensors(const eonerf_ctx* ctx) { return ctx ? (int)ctx->pl.t.size() : 0; }
size_t eonerf_param_floats(const eonerf_ctx* ctx) { return ctx ? ctx->pl.total : 0; }
int eonerf_param_info(const eonerf_ctx* ctx, int index, const char** name, size_t* offset, int* rows, int* cols) {
    if (!ctx || index < 0 || index >= (int)ctx->pl.t.size()) return EONERF_E_ARG;
    const ParamInfo& p = ctx->pl.t[index];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.offset;
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    return EONERF_OK;
}

int eonerf_set_weights(eonerf_ctx* ctx, const float* flat, void* stream) {
    if (!ctx || !flat) return EONERF_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    // pipelined backward: the camera pass reads the heads-only streams + the stage-stationary trunk weights; the density-only streams
    // ride along when something has read them since the last re-pack (shadow pass on: they would be re-packed a few kernels later anyway)
    std::vector<const DevStream*> v;
    v.push_back(&ctx->fwd_full);
    if (ctx->prec == EONERF_F16X3) {}      // forward streams only
    else if (ctx->pipe) { v.push_back(&ctx->bwd_full_heads); v.push_back(&ctx->bwd_rgb_heads); }
    else { v.push_back(&ctx->bwd_full); v.push_back(&ctx->bwd_rgb); }
    if (ctx->pipe) v.push_back(&ctx->pipe_wt);
    const bool with_dens = ctx->dens_used;
    if (with_dens) {      // the same set ensure_density_streams packs
        v.push_back(&ctx->fwd_dens);
        if (ctx->prec != EONERF_F16X3) v.push_back(&ctx->bwd_dens);
        if (ctx->pipe) { v.push_back(&ctx->bwd_dens_heads); v.push_back(&ctx->ig_tail_wt); }
    }
    ctx->need_repack = false;
    int rc = fold_heads(ctx, flat, st);      // the folded head weights are a gather source of the streams below
    if (!rc && ctx->prec == EONERF_F16X3) {      // the weight criteria describe THESE weights until the next re-pack (eonerf_range_status does not clear them)
        rc = (int)hipMemsetAsync(ctx->dev_status + RANGE_STICKY_WORD, 0, sizeof(int), st);
        if (!rc) rc = weight_range(ctx, flat, st);
    }
    if (!rc) rc = pack(ctx, v, flat, st);
    if (!rc) { ctx->weights_set = true; ctx->dens_dirty = !with_dens; ctx->full_ig_dirty = true; ctx->dens_used = false; }
    return rc;
}

size_t eonerf_field_workspace_bytes(const eonerf_ctx* ctx, int n_points) {
    if (!ctx || n_points < 0) return 0;
    Carver c(nullptr);
    PassBuffers b;
    carve_pass(c, b, 1, round_up(std::max(n_points, 1), 256), true, false, false, ctx->bf16 ? 2 : 4);
    return c.off + 256;
}

size_t eonerf_render_workspace_bytes(const eonerf_ctx* ctx, int n_rays, int flags) {
    if (!ctx || n_rays < 0) return 0;
    return carve_render(ctx, nullptr, n_rays, flags).bytes;
}

static int field_common(eonerf_ctx* ctx, const float* flat, const float* xyz, const int64_t* img, int n, bool full,
                        void* ws, size_t ws_bytes, PassBuffers& b, int& p_cap, hipStream_t st) {
    if (!ctx || !xyz || n < 0 || !ws) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (ws_bytes < eonerf_field_workspace_bytes(ctx, n)) return EONERF_E_WORKSPACE;
    drop_presample(ctx, ws);
    p_cap = round_up(std::max(n, 1), 256);
    Carver c(ws);
    carve_pass(c, b, 1, p_cap, true, false, false, ctx->bf16 ? 2 : 4);
    HIP_TRY(eo_launch_points_to_soa(xyz, img, n, p_cap, b.px, b.py, b.pz, b.simg, b.n_pts, st));
    return run_mlp_fwd(ctx, b, flat, p_cap, full, 0, st);
}

int eonerf_field_forward(eonerf_ctx* ctx, const float* flat, const float* xyz, const float* sun, const int64_t* img, int n,
                         float* sigma, float* albedo, float* ambient, float* ts, float* tb,
                         void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return EONERF_OK;
    if (!sun || !img || !sigma || !albedo || !ambient || !ts || !tb || !flat) return EONERF_E_ARG;
    PassBuffers b; int p_cap;
    int rc = field_common(ctx, flat, xyz, img, n, true, ws, ws_bytes, b, p_cap, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(sigma, b.sigma, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(ts, b.ts, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(tb, b.tb, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(eo_launch_soa3_to_aos(b.albedo, p_cap, n, albedo, st));
    return (int)eo_launch_ambient_points(ambient_w(ctx, flat), sun, n, ambient, st);
}

int eonerf_query_density(eonerf_ctx* ctx, const float* flat, const float* xyz, int n, float* sigma, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return EONERF_OK;
    if (!sigma || !flat) return EONERF_E_ARG;
    PassBuffers b; int p_cap;
    int rc = field_common(ctx, flat, xyz, nullptr, n, false, ws, ws_bytes, b, p_cap, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(sigma, b.sigma, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return EONERF_OK;
}

// ---- differentiable EONerfMLP.forward / query_density (radiance_fields/eonerf.py:141-170 under autograd) ----------------
namespace {
struct FieldTrainWs { PassBuffers b; float* m_bott; int* queue; size_t bytes; };
FieldTrainWs carve_field_train(const eonerf_ctx* ctx, void* base, int p_cap, bool full) {
    Carver c(base);
    FieldTrainWs w;
    carve_pass(c, w.b, 1, p_cap, full, true, true, ctx->bf16 ? 2 : 4);
    w.m_bott = c.take<float>(BOTT_SCRATCH_F);
    w.queue = c.take<int>(4);
    w.bytes = c.off + 256;
    return w;
}
}  // namespace

size_t eonerf_field_train_workspace_bytes(const eonerf_ctx* ctx, int n_points, int density_only) {
    if (!ctx || n_points < 0) return 0;
    return carve_field_train(ctx, nullptr, round_up(std::max(n_points, 1), 256), !density_only).bytes;
}

int eonerf_field_forward_train(eonerf_ctx* ctx, const float* flat, const float* xyz, const float* sun, const int64_t* img, int n,
                               int density_only, float* sigma, float* albedo, float* ambient, float* ts, float* tb,
                               void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !xyz || !sigma || n < 0 || !ws) return EONERF_E_ARG;
    if (!density_only && (!sun || !img || !albedo || !ambient || !ts || !tb)) return EONERF_E_ARG;
    drop_presample(ctx, ws);
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (n == 0) return EONERF_OK;
    if (n > (1 << 30)) return EONERF_E_UNSUPPORTED;
    const int p_cap = round_up(n, 256);
    if (!slabs_addressable(ctx, (size_t)p_cap)) return EONERF_E_UNSUPPORTED;
    const bool full = !density_only;
    FieldTrainWs w = carve_field_train(ctx, ws, p_cap, full);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    if (ctx->need_repack) { const int rcr = eonerf_set_weights(ctx, flat, stream); if (rcr) return rcr; }      // (after the fault fallback)
    HIP_TRY(eo_launch_points_to_soa(xyz, full ? img : nullptr, n, p_cap, w.b.px, w.b.py, w.b.pz, w.b.simg, w.b.n_pts, st));
    int rc = run_mlp_fwd(ctx, w.b, flat, p_cap, full, 1, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(sigma, w.b.sigma, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (!full) return EONERF_OK;
    HIP_TRY(hipMemcpyAsync(ts, w.b.ts, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(tb, w.b.tb, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(eo_launch_soa3_to_aos(w.b.albedo, p_cap, n, albedo, st));
    return (int)eo_launch_ambient_points(ambient_w(ctx, flat), sun, n, ambient, st);
}

int eonerf_field_backward(eonerf_ctx* ctx, const float* flat, const float* sun, int n, int density_only,
                          const float* g_sigma, const float* g_albedo, const float* g_ambient, const float* g_ts, const float* g_tb,
                          float* d_flat, float* d_xyz, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !d_flat || n < 0 || !ws) return EONERF_E_ARG;
    if (ctx->prec == EONERF_F16X3) return EONERF_E_UNSUPPORTED;
    if (ctx->pre.valid && ctx->pre.ws == ws) return EONERF_E_STATE;
    if (!density_only && g_ambient && !sun) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (n == 0) return EONERF_OK;
    if (n > (1 << 30)) return EONERF_E_UNSUPPORTED;
    const int p_cap = round_up(n, 256);
    if (!slabs_addressable(ctx, (size_t)p_cap)) return EONERF_E_UNSUPPORTED;
    const bool full = !density_only;
    FieldTrainWs w = carve_field_train(ctx, ws, p_cap, full);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    const ParamLayout& pl = ctx->pl;
    PassBuffers& b = w.b;
    HIP_TRY(eo_launch_field_grads_to_soa(g_sigma, g_albedo, g_ts, g_tb, n, p_cap, b.g_sigma, b.g_albedo, b.g_ts, b.g_tb, st));
    if (full && ctx->full_ig_dirty) {
        const int rc = pack(ctx, {&ctx->bwd_full_ig}, flat, st);
        if (rc) return rc;
        ctx->full_ig_dirty = false;
    }
    if (!full) { const int rc = ensure_density_streams(ctx, flat, st); if (rc) return rc; }
    const DevStream& bs = full ? ctx->bwd_full_ig : ctx->bwd_dens;
    MlpBwdArgs m;
    memset(&m, 0, sizeof(m));
    m.stagger = ctx->stagger;
    m.n_pts = b.n_pts; m.p_pad = p_cap;
    m.stream = bs.data; m.chunks = bs.chunks; m.n_chunks = bs.n_chunks;
    m.sigma = b.sigma; m.albedo = b.albedo; m.ts = b.ts; m.tb = b.tb;
    m.g_sigma = b.g_sigma; m.g_albedo = b.g_albedo; m.g_ts = b.g_ts; m.g_tb = b.g_tb;
    m.masks = b.masks; m.grd = b.grd; m.g_emb = b.g_emb;
    m.px = b.px; m.py = b.py; m.pz = b.pz; m.g_pos = b.g_pos;
    const int tile = ctx->bf16 ? PBf16::TILE : PF32::TILE;
    HIP_TRY(eo_launch_mlp_bwd(m, ctx->bf16, full, true, full, std::min(ctx->n_cu, p_cap / tile), st));
    const int rc = run_weight_gradients(ctx, flat, d_flat, full ? &b : nullptr, true, full ? nullptr : &b, p_cap, w.m_bott, w.queue, st);
    if (rc) return rc;
    if (d_xyz) HIP_TRY(eo_launch_soa3_to_aos(b.g_pos, p_cap, n, d_xyz, st));
    if (!full) return EONERF_OK;
    HIP_TRY(eo_launch_emb_grad_points(b.g_emb, b.simg, n, d_flat + pl.t[pl.emb].offset, st));
    if (g_ambient)
        HIP_TRY(eo_launch_ambient_points_bwd(ambient_w(ctx, flat), sun, g_ambient, n, d_flat + pl.t[pl.am1_w].offset, d_flat + pl.t[pl.am1_b].offset,
                                             d_flat + pl.t[pl.am2_w].offset, d_flat + pl.t[pl.am2_b].offset, st));
    return EONERF_OK;
}

int eonerf_generate_rays(const eonerf_rpc* rpc, const double* cols, const double* rows, long n, int width,
                         double min_alt, double max_alt, int utm_zone, int south,
                         double sun_elevation_deg, double sun_azimuth_deg, const float offset[3], const float scale[3],
                         float* raw8, float* rays, double* geo, void* stream) {
    if (!rpc || n < 0 || (!raw8 && !rays && !geo) || (!cols != !rows) || (!cols && width < 1) || utm_zone < 1 || utm_zone > 60) return EONERF_E_ARG;
    if (rays && (!offset || !scale)) return EONERF_E_ARG;
    if (n == 0) return EONERF_OK;
    static_assert(sizeof(eonerf_rpc) == sizeof(RpcModel), "RPC struct mismatch");
    RayGenArgs a;
    memcpy(&a.rpc, rpc, sizeof(RpcModel));
    // WGS84, Krueger series coefficients (Karney 2011, eq. 35) -- what PROJ's etmerc evaluates
    const double f = 1.0 / 298.257223563, nn = f / (2.0 - f);
    const double n2 = nn * nn, n3 = n2 * nn, n4 = n3 * nn, n5 = n4 * nn, n6 = n5 * nn;
    a.utm.lon0_deg = utm_zone * 6.0 - 183.0;
    a.utm.e = sqrt(f * (2.0 - f));
    a.utm.k0A = 0.9996 * 6378137.0 / (1.0 + nn) * (1.0 + n2 / 4 + n4 / 64 + n6 / 256);
    a.utm.false_north = south ? 10000000.0 : 0.0;
    a.utm.alpha[0] = nn / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    a.utm.alpha[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    a.utm.alpha[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    a.utm.alpha[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    a.utm.alpha[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    a.utm.alpha[5] = 212378941 * n6 / 319334400;
    a.cols = cols; a.rows = rows; a.n = n; a.width = width; a.min_alt = min_alt; a.max_alt = max_alt;
    // get_sun_dirs(90 - elevation, azimuth) -> get_dir_vec_from_el_az (datasets/satellite.py:457,57-63)
    const double d2r = 0.017453292519943295;
    const double el = (90.0 - (90.0 - sun_elevation_deg)) * d2r, az = sun_azimuth_deg * d2r;
    a.sun[0] = -1.0 * (sin(az) * cos(el)); a.sun[1] = -1.0 * (cos(az) * cos(el)); a.sun[2] = -1.0 * sin(el);
    for (int k = 0; k < 3; ++k) { a.offset[k] = offset ? offset[k] : 0.f; a.scale[k] = scale ? scale[k] : 1.f; }
    a.raw8 = raw8; a.rays = rays; a.geo = geo;
    return (int)eo_launch_raygen(a, (hipStream_t)stream);
}

// The camera pass backwards, from the gradient of the per-ray record (w.g_ray): compositing -> heads chain -> [pipelined trunk] -> weight
// gradients (together with the sun pass' remaining jobs, if any) -> embedding table and, with `ambient`, the per-ray ambient head.
// density_only: the pass was a density-only one (render_depth): its gradient flows through the sigma row alone.
static int camera_backward(eonerf_ctx* ctx, const RenderWs& w, const float* flat, const float* rays, const int64_t* img_idx, int n_rays, int p_cap,
                           float* d_flat, bool transient, bool ambient, bool first_pipe, const PassBuffers* sun, bool density_only, hipStream_t st,
                           const unsigned long long* digest = nullptr, bool sun_enc_done = false) {
    const ParamLayout& pl = ctx->pl;
    const int tile = ctx->bf16 ? PBf16::TILE : PF32::TILE;
    const int grid = std::min(ctx->n_cu, p_cap / tile);
    auto dptr = [&](int ti) { return d_flat + pl.t[ti].offset; };
    CompositeBwdArgs cb;
    memset(&cb, 0, sizeof(cb));
    cb.n_samples = ctx->n_samples;
    cb.rays = rays; cb.p_pad = p_cap; cb.n_rays = n_rays; cb.ray_rec = w.ray_rec; cb.g_ray = w.g_ray;
    cb.offsets = w.cam.offsets; cb.counts = w.cam.counts; cb.sigma = w.cam.sigma; cb.delta = w.cam.delta; cb.tmid = w.cam.tmid;
    cb.albedo = w.cam.albedo; cb.ts = w.cam.ts; cb.tb = w.cam.tb;
    cb.g_sigma = w.cam.g_sigma; cb.g_albedo = w.cam.g_albedo; cb.g_ts = w.cam.g_ts; cb.g_tb = w.cam.g_tb;
    cb.depth_only = density_only ? 1 : 0;
    if (sun && !density_only) { cb.sun_offsets = sun->offsets; cb.sun_counts = sun->counts; cb.sun_g_pos = sun->g_pos; }      // d depth of the shadow rays' origins
    if (digest) { cb.chk_a = digest; cb.chk_b = digest + 1; cb.chk_status = ctx->dev_status; }      // presample guard (eonerf_rays.h)
    HIP_TRY(eo_launch_cam_composite_bwd(cb, st));
    MlpBwdArgs mc;
    memset(&mc, 0, sizeof(mc));
    mc.stagger = ctx->stagger;
    mc.n_pts = w.cam.n_pts; mc.p_pad = p_cap;
    const bool pipe = ctx->pipe && w.pipe.dy_in;
    if (density_only) { const int rc = ensure_density_streams(ctx, flat, st); if (rc) return rc; }
    const DevStream& bs = density_only ? (pipe ? ctx->bwd_dens_heads : ctx->bwd_dens)
                        : pipe ? (transient ? ctx->bwd_full_heads : ctx->bwd_rgb_heads) : (transient ? ctx->bwd_full : ctx->bwd_rgb);
    mc.stream = bs.data; mc.chunks = bs.chunks; mc.n_chunks = bs.n_chunks;
    mc.sigma = w.cam.sigma; mc.albedo = w.cam.albedo; mc.ts = w.cam.ts; mc.tb = w.cam.tb;
    mc.g_sigma = w.cam.g_sigma; mc.g_albedo = w.cam.g_albedo; mc.g_ts = w.cam.g_ts; mc.g_tb = w.cam.g_tb;
    mc.masks = w.cam.masks; mc.grd = w.cam.grd; mc.g_emb = w.cam.g_emb;
    mc.px = w.cam.px; mc.py = w.cam.py; mc.pz = w.cam.pz; mc.g_pos = w.cam.g_pos;      // (density variants only: the chain + GEMM path ends in d position)
    mc.dy7_units = pipe ? w.pipe.dy_in : nullptr;
    { ProfScope ps(ctx, EONERF_PROF_BWD_CHAIN_CAMERA, st);
      HIP_TRY(eo_launch_mlp_bwd(mc, ctx->bf16, !density_only, density_only, transient && !density_only, grid, st, pipe ? 1 : 0)); }
    BottWgradArgs bott;
    const bool stream = pipe && !density_only && ctx->stream_blocks > 0 && !ctx->deterministic && !w.det.wgrad_part && !w.det.pipe_part;
    if (stream) {
        // the GEMM's job table first: the pipelined launch hands its ready items to streaming workgroups (PipeStreamArgs), the GEMM launch
        // behind it takes the rest of the SAME queue (zeroed with the backward's sync block)
        WgradPlan plan;
        const int rcw = run_weight_gradients(ctx, flat, d_flat, &w.cam, transient, sun, p_cap, w.m_bott, w.queue, st, true, sun != nullptr, nullptr, true, &bott, &plan, sun_enc_done);
        if (rcw) return rcw;
        const int rcp = run_bwd_pipe(ctx, w, w.cam, p_cap, d_flat, EONERF_PROF_BWD_PIPE_CAMERA, st, 1, first_pipe, nullptr, &plan);
        if (rcp) return rcp;
        const int rcl = launch_planned_wgrad(ctx, plan, p_cap, w.queue, st);
        if (rcl) return rcl;
    } else {
        if (pipe) {
            const int rcp = run_bwd_pipe(ctx, w, w.cam, p_cap, d_flat, EONERF_PROF_BWD_PIPE_CAMERA, st, 1, first_pipe); if (rcp) return rcp;
            // the trunk's pipelined layers are complete in d_flat here (the shadow pass' launch ran before this one)
            if (ctx->exch_event && !density_only) { HIP_TRY(hipEventRecord(ctx->exch_event, st)); ctx->exch_recorded = true; }
        }
        const PassBuffers* full = density_only ? nullptr : &w.cam;
        const PassBuffers* dens = density_only ? &w.cam : sun;
        const int rcw = run_weight_gradients(ctx, flat, d_flat, full, transient, dens, p_cap, w.m_bott, w.queue, st, pipe, pipe && dens, w.det.wgrad_part, pipe,
                                             (full && !ctx->deterministic) ? &bott : nullptr, nullptr, sun_enc_done && !density_only);
        if (rcw) return rcw;
    }
    if (density_only) return EONERF_OK;
    if (!ctx->deterministic) {
        // the three independent tails of the backward -- bottleneck-factor products, embedding table, per-ray ambient head -- in ONE launch
        EmbGradArgs eg;
        eg.n_samples = ctx->n_samples;
        eg.offsets = w.cam.offsets; eg.counts = w.cam.counts; eg.img_idx = img_idx; eg.g_emb = w.cam.g_emb; eg.d_emb = dptr(pl.emb); eg.n_rays = n_rays;
        eg.lds_images = ctx->cfg.n_images <= 4096 ? ctx->cfg.n_images : 0; eg.d_emb_rays = nullptr;
        AmbientBwdArgs ag;
        ag.w = ambient_w(ctx, flat); ag.rays = rays; ag.ray_rec = w.ray_rec; ag.g_ray = w.g_ray; ag.amb_save = w.amb_save; ag.n_rays = n_rays;
        ag.d_w1 = dptr(pl.am1_w); ag.d_b1 = dptr(pl.am1_b); ag.d_w2 = dptr(pl.am2_w); ag.d_b2 = dptr(pl.am2_b);
        EncPartReduceArgs er;
        er.part = w.enc_part; er.n_wg = ctx->n_cu; er.dw0 = dptr(pl.trunk_w[0]); er.db0 = dptr(pl.trunk_b[0]); er.dw5s = dptr(pl.trunk_w[5]) + 256; er.col_map = ctx->enc_colmap;
        return (int)eo_launch_step_tail(&bott, transient ? &eg : nullptr, ambient ? &ag : nullptr, st, sun_enc_done ? &er : nullptr);
    }

    // ---- embeddings and the per-ray ambient head -----------------------------------------------------------
    if (transient) {
        EmbGradArgs eg;
        eg.n_samples = ctx->n_samples;
        eg.offsets = w.cam.offsets; eg.counts = w.cam.counts; eg.img_idx = img_idx; eg.g_emb = w.cam.g_emb; eg.d_emb = dptr(pl.emb); eg.n_rays = n_rays; eg.lds_images = ctx->cfg.n_images <= 4096 ? ctx->cfg.n_images : 0;
        eg.d_emb_rays = w.det.emb_rays;
        HIP_TRY(eo_launch_emb_grad(eg, st));
        if (eg.d_emb_rays) HIP_TRY(eo_launch_table_reduce(eg.d_emb_rays, img_idx, n_rays, 4, 4, ctx->cfg.n_images, 0, eg.d_emb, st));
    }
    if (!ambient) return EONERF_OK;      // s == 1: rgb = albedo, the ambient head is outside the graph (sat_rendering.py:269-276,294)
    // (27 -> 128 -> 3, fp32, ~35 us on a few dozen workgroups.  Running it on a side stream beside the weight-gradient GEMM was tried
    //  and bought nothing: every large kernel of the step holds the whole register file of its CUs -- 8 waves x 256 registers -- so
    //  the small kernel's workgroups only start when the large one's leave)
    AmbientBwdArgs ag;
    ag.w = ambient_w(ctx, flat); ag.rays = rays; ag.ray_rec = w.ray_rec; ag.g_ray = w.g_ray; ag.amb_save = w.amb_save; ag.n_rays = n_rays;
    ag.d_w1 = dptr(pl.am1_w); ag.d_b1 = dptr(pl.am1_b); ag.d_w2 = dptr(pl.am2_w); ag.d_b2 = dptr(pl.am2_b);
    HIP_TRY(eo_launch_ambient_bwd(ag, st, ctx->deterministic));
    return EONERF_OK;
}

int eonerf_set_n_samples(eonerf_ctx* ctx, int n_samples) {
    if (!ctx) return EONERF_E_ARG;
    if (n_samples < 2 || n_samples > 256) return EONERF_E_UNSUPPORTED;
    ctx->n_samples = n_samples;
    return EONERF_OK;
}

int eonerf_set_noise_seed(eonerf_ctx* ctx, uint64_t seed) {
    if (!ctx) return EONERF_E_ARG;
    ctx->noise_seed = seed; ctx->noise_call = 0; ctx->pre.valid = false;
    return EONERF_OK;
}

int eonerf_sample_rays(eonerf_ctx* ctx, const float* rays, const float* zsteps, const float* u, int perturb, int n_rays,
                       int64_t* ray_indices, float* t_starts, float* t_ends, float* pts_per_ray, int* n_dev,
                       void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_rays == 0) return EONERF_OK;
    if (!ctx || !rays || !zsteps || !ray_indices || !t_starts || !t_ends || n_rays < 0 || !ws) return EONERF_E_ARG;
    RenderWs w = carve_render(ctx, ws, n_rays, EONERF_F_ONLY_DEPTH);
    drop_presample(ctx, ws);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    SampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n_samples = ctx->n_samples;
    sa.rays = rays; sa.zsteps = zsteps; sa.u = u; sa.n_rays = n_rays; sa.perturb = perturb ? 1 : 0;
    if (perturb && !u) { sa.seed = ctx->noise_seed; sa.call = ctx->noise_call++; }
    sa.cnt_first = w.cnt_first; sa.cnt_retry = w.cnt_retry; sa.counts = w.cam.counts; sa.offsets = w.cam.offsets;
    sa.flags = w.flags; sa.n_pts = w.cam.n_pts;
    sa.px = w.cam.px; sa.py = w.cam.py; sa.pz = w.cam.pz; sa.tmid = w.cam.tmid; sa.delta = w.cam.delta; sa.simg = w.cam.simg;
    sa.o_ray = ray_indices; sa.o_ts = t_starts; sa.o_te = t_ends;
    HIP_TRY(eo_launch_sampler(sa, st));
    if (n_dev) HIP_TRY(hipMemcpyAsync(n_dev, w.cam.n_pts, sizeof(int), hipMemcpyDeviceToDevice, st));
    if (pts_per_ray) HIP_TRY(eo_launch_int_to_float(w.cam.counts, n_rays, pts_per_ray, st));
    return EONERF_OK;
}

int eonerf_rendering(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                     const float* t_starts, const float* t_ends, const int64_t* ray_indices, int n, int n_rays, int depth_only,
                     float* albedo, float* depth, float* beta, float* transient_s, float* ambient, float* entropy,
                     void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !rays || !depth || n < 0 || n_rays < 1 || !ws) return EONERF_E_ARG;
    if (n > 0 && (!t_starts || !t_ends || !ray_indices)) return EONERF_E_ARG;
    if (!depth_only && (!albedo || !beta || !transient_s || !ambient || !entropy || !img_idx)) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (!rays_in_range(ctx, n_rays) || (long long)n > (long long)n_rays * (ctx->n_samples - 1)) return EONERF_E_UNSUPPORTED;      // at most n_samples - 1 intervals per ray
    const int flags = depth_only ? EONERF_F_ONLY_DEPTH : 0;
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    drop_presample(ctx, ws);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);
    PackedArgs pa;
    pa.rays = rays; pa.img_idx = img_idx; pa.t_starts = t_starts; pa.t_ends = t_ends; pa.ray_indices = ray_indices;
    pa.n = n; pa.n_rays = n_rays; pa.counts = w.cam.counts; pa.offsets = w.cam.offsets; pa.n_pts = w.cam.n_pts;
    pa.px = w.cam.px; pa.py = w.cam.py; pa.pz = w.cam.pz; pa.tmid = w.cam.tmid; pa.delta = w.cam.delta; pa.simg = w.cam.simg;
    HIP_TRY(eo_launch_from_packed(pa, st));
    int rc = run_mlp_fwd(ctx, w.cam, flat, p_cap, !depth_only, 0, st);
    if (rc) return rc;
    CompositeArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.n_samples = ctx->n_samples;
    ca.rays = rays; ca.offsets = w.cam.offsets; ca.counts = w.cam.counts;
    ca.sigma = w.cam.sigma; ca.delta = w.cam.delta; ca.tmid = w.cam.tmid; ca.albedo = w.cam.albedo; ca.ts = w.cam.ts; ca.tb = w.cam.tb;
    ca.p_pad = p_cap; ca.n_rays = n_rays; ca.depth_only = depth_only ? 1 : 0; ca.amb = ambient_w(ctx, flat); ca.ray_out = w.ray_rec;
    HIP_TRY(eo_launch_composite_fwd(ca, st));
    RenderingOutArgs ro{w.ray_rec, n_rays, depth_only ? nullptr : albedo, depth, beta, transient_s, ambient, entropy};
    return (int)eo_launch_rendering_out(ro, st);
}

// EONerfMLP.rendering / render_depth under autograd (radiance_fields/eonerf.py:172-248): the same kernels with the training-mode
// forward chain (activations, masks and the compositing inputs stay in the workspace for eonerf_rendering_backward)
int eonerf_rendering_train(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                           const float* t_starts, const float* t_ends, const int64_t* ray_indices, int n, int n_rays, int depth_only,
                           float* albedo, float* depth, float* beta, float* transient_s, float* ambient, float* entropy,
                           void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !rays || !depth || n < 0 || n_rays < 1 || !ws) return EONERF_E_ARG;
    if (n > 0 && (!t_starts || !t_ends || !ray_indices)) return EONERF_E_ARG;
    if (!depth_only && (!albedo || !beta || !transient_s || !ambient || !entropy || !img_idx)) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (!rays_in_range(ctx, n_rays) || !slabs_addressable(ctx, (size_t)p_cap_of(n_rays, ctx->n_samples))) return EONERF_E_UNSUPPORTED;
    if ((long long)n > (long long)n_rays * (ctx->n_samples - 1)) return EONERF_E_UNSUPPORTED;           // at most n_samples - 1 intervals per ray
    if (ctx->need_repack) { const int rcr = eonerf_set_weights(ctx, flat, stream); if (rcr) return rcr; }      // (after the fault fallback)
    const int flags = EONERF_F_TRAIN | (depth_only ? EONERF_F_ONLY_DEPTH : 0);
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    drop_presample(ctx, ws);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    note_train_forward(ctx, ws);
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);
    PackedArgs pa;
    pa.rays = rays; pa.img_idx = img_idx; pa.t_starts = t_starts; pa.t_ends = t_ends; pa.ray_indices = ray_indices;
    pa.n = n; pa.n_rays = n_rays; pa.counts = w.cam.counts; pa.offsets = w.cam.offsets; pa.n_pts = w.cam.n_pts;
    pa.px = w.cam.px; pa.py = w.cam.py; pa.pz = w.cam.pz; pa.tmid = w.cam.tmid; pa.delta = w.cam.delta; pa.simg = w.cam.simg;
    HIP_TRY(eo_launch_from_packed(pa, st));
    int rc = run_mlp_fwd(ctx, w.cam, flat, p_cap, !depth_only, 1, st, -1, true);
    if (rc) return rc;
    CompositeArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.n_samples = ctx->n_samples;
    ca.rays = rays; ca.offsets = w.cam.offsets; ca.counts = w.cam.counts;
    ca.sigma = w.cam.sigma; ca.delta = w.cam.delta; ca.tmid = w.cam.tmid; ca.albedo = w.cam.albedo; ca.ts = w.cam.ts; ca.tb = w.cam.tb;
    ca.p_pad = p_cap; ca.n_rays = n_rays; ca.depth_only = depth_only ? 1 : 0; ca.amb = ambient_w(ctx, flat); ca.ray_out = w.ray_rec;
    ca.amb_save = depth_only ? nullptr : w.amb_save;
    HIP_TRY(eo_launch_composite_fwd(ca, st));
    RenderingOutArgs ro{w.ray_rec, n_rays, depth_only ? nullptr : albedo, depth, beta, transient_s, ambient, entropy};
    return (int)eo_launch_rendering_out(ro, st);
}

// gradients of the per-ray outputs (row-major as autograd hands them over; null = zero) -> ACCUMULATED parameter gradients.  `ws` must be
// the workspace an eonerf_rendering_train call with the same (rays, img_idx, n_rays, depth_only) filled.
int eonerf_rendering_backward(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx, int n_rays, int depth_only,
                              const float* g_albedo, const float* g_depth, const float* g_beta, const float* g_transient_s, const float* g_ambient,
                              float* d_flat, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !rays || !d_flat || n_rays < 1 || !ws) return EONERF_E_ARG;
    if (ctx->prec == EONERF_F16X3) return EONERF_E_UNSUPPORTED;
    if (!depth_only && !img_idx) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (!rays_in_range(ctx, n_rays) || !slabs_addressable(ctx, (size_t)p_cap_of(n_rays, ctx->n_samples))) return EONERF_E_UNSUPPORTED;      // (the forward's own bounds)
    PipeModeGuard mode(ctx, ws);
    const int flags = EONERF_F_TRAIN | (depth_only ? EONERF_F_ONLY_DEPTH : 0);
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    if (ctx->pre.valid && ctx->pre.ws == ws) return EONERF_E_STATE;      // eonerf_presample ran between this backward and its forward
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);
    RenderingOutBwdArgs rb{w.ray_rec, n_rays, depth_only ? nullptr : g_albedo, g_depth, depth_only ? nullptr : g_beta,
                           depth_only ? nullptr : g_transient_s, depth_only ? nullptr : g_ambient, w.g_ray};
    HIP_TRY(eo_launch_rendering_out_bwd(rb, st));
    return camera_backward(ctx, w, flat, rays, img_idx, n_rays, p_cap, d_flat, true, !depth_only, true, nullptr, depth_only != 0, st);
}

// Arguments of the camera pass's sampler launch (eonerf_render_forward, eonerf_presample); the Philox call number is the caller's
static SampleArgs camera_sample_args(const eonerf_ctx* ctx, const RenderWs& w, const float* rays, const int64_t* img_idx, const float* zsteps,
                                     const float* u_cam, const float* u_retry, int n_rays, int* n_samples_dev) {
    SampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.n_samples = ctx->n_samples;
    sa.rays = rays; sa.img_idx = img_idx; sa.zsteps = zsteps; sa.u = u_cam; sa.u_retry = u_retry;
    sa.perturb = 1; sa.retry = (!u_cam || u_retry) ? 1 : 0;
    if (!u_cam) sa.seed = ctx->noise_seed;
    sa.n_rays = n_rays; sa.sun_pass = 0; sa.patch_last = 1;
    sa.cnt_first = w.cnt_first; sa.cnt_retry = w.cnt_retry; sa.counts = w.cam.counts; sa.offsets = w.cam.offsets;
    sa.flags = w.flags; sa.n_pts = w.cam.n_pts; sa.n_pts_copy = n_samples_dev;       // the scan kernel also fills the caller's count
    sa.px = w.cam.px; sa.py = w.cam.py; sa.pz = w.cam.pz; sa.tmid = w.cam.tmid; sa.delta = w.cam.delta; sa.simg = w.cam.simg;
    return sa;
}

/* The camera pass's sampler of the NEXT eonerf_render_forward(EONERF_F_TRAIN, production noise), launched ahead of it: it reads the rays and
 * the seed only, so a data-parallel trainer runs it on the compute stream while the gradient all-reduce of the step before is in flight
 * (the exchange's serial tail).  The workspace must be free (the backward that used it has been enqueued on `stream`). */
int eonerf_presample(eonerf_ctx* ctx, const float* rays, const int64_t* img_idx, const float* zsteps, int n_rays, int flags,
                     int* n_samples_dev, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !rays || !img_idx || !zsteps || n_rays < 0 || !ws) return EONERF_E_ARG;
    if (!(flags & EONERF_F_TRAIN) || (flags & EONERF_F_ONLY_DEPTH)) return EONERF_E_STATE;
    ctx->pre.valid = false;
    if (n_rays == 0) return EONERF_OK;
    if (!rays_in_range(ctx, n_rays) || !slabs_addressable(ctx, (size_t)p_cap_of(n_rays, ctx->n_samples))) return EONERF_E_UNSUPPORTED;
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    SampleArgs sa = camera_sample_args(ctx, w, rays, img_idx, zsteps, nullptr, nullptr, n_rays, n_samples_dev);
    sa.call = ctx->noise_call++;
    // content guard: the sampler sums a digest of the rays it reads; the backward of the forward that consumes the record sums it again
    // from the same buffers and raises the status word if they were refilled in between (pointer identity alone cannot see that)
    unsigned long long* digest = reinterpret_cast<unsigned long long*>(ctx->dev_status + DIGEST_WORD);
    HIP_TRY(hipMemsetAsync(digest, 0, 2 * sizeof(unsigned long long), st));
    sa.digest = digest;
    HIP_TRY(eo_launch_sampler(sa, st));
    eonerf_ctx::Presample& p = ctx->pre;
    p.valid = true; p.ws = ws; p.rays = rays; p.img_idx = img_idx; p.zsteps = zsteps; p.count_out = n_samples_dev; p.n_rays = n_rays; p.flags = flags;
    p.n_samples = ctx->n_samples; p.pipe = ctx->pipe; p.call = sa.call;
    return EONERF_OK;
}

int eonerf_presample_cancel(eonerf_ctx* ctx) {
    if (!ctx) return EONERF_E_ARG;
    ctx->pre.valid = false;
    return EONERF_OK;
}

int eonerf_render_forward(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                          const float* zsteps, const float* u_cam, const float* u_retry, const float* u_sun,
                          int n_rays, int flags, float* out, int* n_samples_dev,
                          void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !rays || !img_idx || !zsteps || !out || n_rays < 0 || !ws) return EONERF_E_ARG;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (n_rays == 0) return EONERF_OK;
    if (ctx->need_repack) { const int rcr = eonerf_set_weights(ctx, flat, stream); if (rcr) return rcr; }
    const bool shadows = (flags & EONERF_F_SHADOWS) && !(flags & EONERF_F_ONLY_DEPTH);
    const bool train = flags & EONERF_F_TRAIN, od = flags & EONERF_F_ONLY_DEPTH;
    const bool philox = u_cam == nullptr;       // production: no noise buffers, the sampler draws its own jitter
    if (philox ? (u_retry || u_sun) : (shadows && !u_sun)) return EONERF_E_ARG;
    if (train && od) return EONERF_E_UNSUPPORTED;
    if (!rays_in_range(ctx, n_rays) || (train && !slabs_addressable(ctx, (size_t)p_cap_of(n_rays, ctx->n_samples)))) return EONERF_E_UNSUPPORTED;
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    if (train) note_train_forward(ctx, ws);
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);

    // ---- camera pass: sample -> field -> composite -------------------------------------------------------
    const eonerf_ctx::Presample pre = ctx->pre;
    const bool presampled = pre.valid && philox && pre.ws == ws && pre.rays == rays && pre.img_idx == img_idx && pre.zsteps == zsteps &&
                            pre.count_out == n_samples_dev && pre.n_rays == n_rays && pre.flags == flags && pre.n_samples == ctx->n_samples && pre.pipe == ctx->pipe;
    ctx->pre.valid = false;      // consumed, or dropped: this call's kernels write the workspace the record described (or the caller moved on)
    ctx->pre_consumed_ws = (presampled && train) ? ws : nullptr;
    SampleArgs sa = camera_sample_args(ctx, w, rays, img_idx, zsteps, u_cam, u_retry, n_rays, n_samples_dev);
    if (presampled) sa.call = pre.call;                       // (the shadow pass draws under the same call number)
    else {
        if (philox) sa.call = ctx->noise_call++;
        HIP_TRY(eo_launch_sampler(sa, st));
    }
    const bool rgb_loss = train && !shadows && (flags & EONERF_F_RGB_LOSS);
    int rc = run_mlp_fwd(ctx, w.cam, flat, p_cap, !od, train ? (rgb_loss ? 2 : 1) : 0, st, EONERF_PROF_FWD_CHAIN_CAMERA, train);
    if (rc) return rc;
    CompositeArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.n_samples = ctx->n_samples;
    ca.rays = rays; ca.offsets = w.cam.offsets; ca.counts = w.cam.counts;
    ca.sigma = w.cam.sigma; ca.delta = w.cam.delta; ca.tmid = w.cam.tmid; ca.albedo = w.cam.albedo; ca.ts = w.cam.ts; ca.tb = w.cam.tb;
    ca.p_pad = p_cap; ca.n_rays = n_rays; ca.shadow_only = 0; ca.depth_only = od ? 1 : 0;
    ca.amb = ambient_w(ctx, flat); ca.ray_out = w.ray_rec; ca.amb_save = w.amb_save;
    // irradiance model + radiometric affine + packing (sat_rendering.py:265-312): done by the chunk's LAST compositing launch, ray by ray
    ShadeArgs sh;
    sh.ray_rec = w.ray_rec; sh.img_idx = img_idx;
    sh.radiometric = ctx->cfg.radiometric ? flat + ctx->pl.t[ctx->pl.rad].offset : nullptr;
    sh.pts_first = w.cnt_first; sh.sc_counts = shadows ? w.sun.counts : w.cnt_first;
    sh.n_rays = n_rays; sh.use_shadow = shadows ? 1 : 0; sh.eval = (flags & EONERF_F_EVAL) ? 1 : 0; sh.out = out;
    ca.shade = sh; ca.do_shade = shadows ? 0 : 1;
    // sun pass: shadow rays from the rendered surface toward the sun; the camera compositing counts their samples
    SampleArgs ss = sa;
    ss.img_idx = nullptr; ss.u = u_sun; ss.u_retry = nullptr; ss.retry = 0;
    ss.depth = w.ray_rec + RR_DEPTH; ss.depth_stride = RAY_REC; ss.sun_pass = 1; ss.patch_last = 0;
    ss.cnt_first = w.sun.counts; ss.cnt_retry = w.cnt_retry; ss.counts = w.sun.counts; ss.offsets = w.sun.offsets;
    ss.n_pts = w.sun.n_pts; ss.n_pts_copy = nullptr;
    ss.px = w.sun.px; ss.py = w.sun.py; ss.pz = w.sun.pz; ss.tmid = w.sun.tmid; ss.delta = w.sun.delta; ss.simg = w.sun.simg;
    if (shadows) { ca.count_sun = 1; ca.sun = ss; }
    HIP_TRY(eo_launch_composite_fwd(ca, st));

    // ---- sun pass ---------------------------------------------------------------------------------------------
    if (shadows) {
        HIP_TRY(eo_launch_sampler(ss, st, true));
        rc = run_mlp_fwd(ctx, w.sun, flat, p_cap, false, train ? 1 : 0, st, EONERF_PROF_FWD_CHAIN_SUN, train);
        if (rc) return rc;
        CompositeArgs cs = ca;
        cs.offsets = w.sun.offsets; cs.counts = w.sun.counts; cs.sigma = w.sun.sigma; cs.delta = w.sun.delta; cs.tmid = w.sun.tmid;
        cs.shadow_only = 1; cs.depth_only = 0; cs.count_sun = 0; cs.do_shade = 1;
        HIP_TRY(eo_launch_composite_fwd(cs, st));
    }
    return EONERF_OK;
}

struct LossSpec { const float *out, *pixels; int kind; float* loss; };      // fused loss (eonerf_render_backward_loss) or nullptr
static int render_backward_impl(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                                int n_rays, int flags, const float* d_out, const LossSpec* ls, float* d_flat,
                                void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!ctx || !flat || !rays || !img_idx || (!d_out && !ls) || !d_flat || n_rays < 0 || !ws) return EONERF_E_ARG;
    if (!(flags & EONERF_F_TRAIN) || (flags & EONERF_F_ONLY_DEPTH)) return EONERF_E_STATE;
    if (ctx->prec == EONERF_F16X3) return EONERF_E_UNSUPPORTED;
    if (!ctx->weights_set) return EONERF_E_STATE;
    if (n_rays == 0) return EONERF_OK;
    if (!rays_in_range(ctx, n_rays) || !slabs_addressable(ctx, (size_t)p_cap_of(n_rays, ctx->n_samples))) return EONERF_E_UNSUPPORTED;
    PipeModeGuard mode(ctx, ws);
    const bool shadows = flags & EONERF_F_SHADOWS;
    ctx->exch_recorded = false;
    RenderWs w = carve_render(ctx, ws, n_rays, flags);
    if (ctx->pre.valid && ctx->pre.ws == ws) return EONERF_E_STATE;      // eonerf_presample ran between this backward and its forward
    if (ws_bytes < w.bytes) return EONERF_E_WORKSPACE;
    const int p_cap = p_cap_of(n_rays, ctx->n_samples);
    const ParamLayout& pl = ctx->pl;
    const int tile = ctx->bf16 ? PBf16::TILE : PF32::TILE;
    const int grid = std::min(ctx->n_cu, p_cap / tile);
    auto dptr = [&](int ti) { return d_flat + pl.t[ti].offset; };

    // ---- d out -> d ray record (+ radiometric table) -----------------------------------------------------
    ShadeBwdArgs sb;
    sb.ray_rec = w.ray_rec; sb.d_out = d_out; sb.img_idx = img_idx;
    sb.loss_kind = -1; sb.loss_out = nullptr; sb.loss_gt = nullptr; sb.loss = nullptr; sb.loss_scratch = nullptr;
    sb.chk_rays = nullptr; sb.chk_sum = nullptr;
    if (ls) { sb.loss_kind = ls->kind; sb.loss_out = ls->out; sb.loss_gt = ls->pixels; sb.loss = ls->loss; sb.loss_scratch = ctx->loss_scratch; }
    sb.radiometric = ctx->cfg.radiometric ? flat + pl.t[pl.rad].offset : nullptr;
    sb.d_radiometric = ctx->cfg.radiometric ? dptr(pl.rad) : nullptr;
    sb.g_ray = w.g_ray; sb.n_rays = n_rays; sb.use_shadow = shadows ? 1 : 0; sb.eval = (flags & EONERF_F_EVAL) ? 1 : 0;
    sb.lds_images = ctx->cfg.n_images <= 2048 ? ctx->cfg.n_images : 0;
    sb.d_rad_rays = sb.d_radiometric ? w.det.rad_rays : nullptr;
    const bool chk = ctx->pre_consumed_ws == ws;      // this backward's forward ran on presampled rays: digest of the buffers as they are now
    ctx->pre_consumed_ws = nullptr;
    unsigned long long* digest = reinterpret_cast<unsigned long long*>(ctx->dev_status + DIGEST_WORD);
    sb.chk_rays = chk ? rays : nullptr; sb.chk_sum = chk ? digest + 1 : nullptr;
    // pipelined path: this first kernel of the call also zeroes [bottleneck factors | GEMM queue | sync blocks] (pipe_clear's memset)
    const bool prezeroed = ctx->pipe && w.pipe.dy_in;
    sb.zero_base = nullptr; sb.zero_bytes = 0;
    if (prezeroed) {
        uint8_t* lo = reinterpret_cast<uint8_t*>(w.m_bott);
        uint8_t* hi = reinterpret_cast<uint8_t*>(w.pipe.sync) + PIPE_LAUNCHES * w.pipe.sync_bytes;
        sb.zero_base = reinterpret_cast<uint32_t*>(lo); sb.zero_bytes = (size_t)(hi - lo);
        if ((reinterpret_cast<uintptr_t>(lo) | sb.zero_bytes) & 15) return EONERF_E_STATE;
    }
    HIP_TRY(eo_launch_shade_bwd(sb, st));
    if (sb.d_rad_rays) HIP_TRY(eo_launch_table_reduce(sb.d_rad_rays, img_idx, n_rays, 6, 9, ctx->cfg.n_images, sb.eval, sb.d_radiometric, st));

    CompositeBwdArgs cb;
    memset(&cb, 0, sizeof(cb));
    cb.n_samples = ctx->n_samples;
    cb.rays = rays; cb.p_pad = p_cap; cb.n_rays = n_rays; cb.ray_rec = w.ray_rec; cb.g_ray = w.g_ray;
    bool ambient_done = false, sun_enc_done = false;

    // ---- shadow pass backwards: d geo -> d sigma_sun -> (chain, input grad) -> d pos -> d depth -----------
    if (shadows) {
        CompositeBwdArgs cs = cb;
        cs.offsets = w.sun.offsets; cs.counts = w.sun.counts; cs.sigma = w.sun.sigma; cs.delta = w.sun.delta;
        cs.g_sigma = w.sun.g_sigma; cs.g_pos = w.sun.g_pos;
        HIP_TRY(eo_launch_sun_composite_bwd(cs, st));
        MlpBwdArgs ms;
        memset(&ms, 0, sizeof(ms));
    ms.stagger = ctx->stagger;
        ms.n_pts = w.sun.n_pts; ms.p_pad = p_cap;
        ms.stream = ctx->bwd_dens.data; ms.chunks = ctx->bwd_dens.chunks; ms.n_chunks = ctx->bwd_dens.n_chunks;
        ms.sigma = w.sun.sigma; ms.g_sigma = w.sun.g_sigma; ms.masks = w.sun.masks; ms.grd = w.sun.grd;
        ms.px = w.sun.px; ms.py = w.sun.py; ms.pz = w.sun.pz; ms.g_pos = w.sun.g_pos;
        const bool pipe_sun = ctx->pipe && w.pipe.dy_in;
        if (pipe_sun) {      // heads (sigma row) -> pipelined trunk -> input-gradient tail
            ms.stream = ctx->bwd_dens_heads.data; ms.chunks = ctx->bwd_dens_heads.chunks; ms.n_chunks = ctx->bwd_dens_heads.n_chunks;
            ms.dy7_units = w.pipe.dy_in;
            { ProfScope ps(ctx, EONERF_PROF_BWD_CHAIN_SUN, st); HIP_TRY(eo_launch_mlp_bwd(ms, true, false, true, false, grid, st, true)); }
            // (the spare CUs of this launch take the ambient-head backward: its inputs -- g_ray, the saved head activations -- are final)
            AmbientBwdArgs ag;
            ag.w = ambient_w(ctx, flat); ag.rays = rays; ag.ray_rec = w.ray_rec; ag.g_ray = w.g_ray; ag.amb_save = w.amb_save; ag.n_rays = n_rays;
            ag.d_w1 = dptr(pl.am1_w); ag.d_b1 = dptr(pl.am1_b); ag.d_w2 = dptr(pl.am2_w); ag.d_b2 = dptr(pl.am2_b);
            ambient_done = pipe_spare_cus(ctx) > 0;
            const int rcp = run_bwd_pipe(ctx, w, w.sun, p_cap, d_flat, EONERF_PROF_BWD_PIPE_SUN, st, 0, !prezeroed, ambient_done ? &ag : nullptr);
            if (rcp) return rcp;
            if (w.enc_part) {
                // ONE pass over the dY_0 / dY_5 tiles the launch above left: d sigma / d position (needed now: it flows into the camera pass)
                // and the pass' two weight-gradient products against the encoding (otherwise two jobs of the GEMM launch at the end)
                EncPairArgs ea;
                ea.n_pts = w.sun.n_pts; ea.p_pad = p_cap; ea.grd = w.sun.grd; ea.act = w.sun.act; ea.wt = ctx->ig_tail_wt.data;
                ea.px = w.sun.px; ea.py = w.sun.py; ea.pz = w.sun.pz; ea.g_pos = w.sun.g_pos;
                ea.part = w.enc_part;
                { ProfScope ps(ctx, EONERF_PROF_IG_TAIL_SUN, st); HIP_TRY(eo_launch_enc_pair(ea, ctx->n_cu, st)); }
                sun_enc_done = true;
            } else {
            IgTailArgs ta;
            ta.n_pts = w.sun.n_pts; ta.p_pad = p_cap; ta.grd = w.sun.grd; ta.wt = ctx->ig_tail_wt.data;
            ta.px = w.sun.px; ta.py = w.sun.py; ta.pz = w.sun.pz; ta.g_pos = w.sun.g_pos;
            { ProfScope ps(ctx, EONERF_PROF_IG_TAIL_SUN, st); HIP_TRY(eo_launch_ig_tail(ta, ctx->n_cu, st)); }
            }
        } else {
            ProfScope ps(ctx, EONERF_PROF_BWD_CHAIN_SUN, st);
            HIP_TRY(eo_launch_mlp_bwd(ms, ctx->bf16, false, true, false, grid, st));
        }
    }

    const int rcc = camera_backward(ctx, w, flat, rays, img_idx, n_rays, p_cap, d_flat, shadows || !(flags & EONERF_F_RGB_LOSS), shadows && !ambient_done, !shadows && !prezeroed,
                                    shadows ? &w.sun : nullptr, false, st, chk ? digest : nullptr, sun_enc_done);
    // (chain + GEMM path: the trunk's gradients come out of the GEMM launch -- the early block is final where everything is)
    if (!rcc && ctx->exch_event && !ctx->exch_recorded) HIP_TRY(hipEventRecord(ctx->exch_event, st));
    return rcc;
}

int eonerf_render_backward(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                           int n_rays, int flags, const float* d_out, float* d_flat,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!d_out) return EONERF_E_ARG;
    return render_backward_impl(ctx, flat, rays, img_idx, n_rays, flags, d_out, nullptr, d_flat, ws, ws_bytes, stream);
}

int eonerf_render_backward_loss(eonerf_ctx* ctx, const float* flat, const float* rays, const int64_t* img_idx,
                                int n_rays, int flags, const float* out, const float* pixels, int kind, float* d_out_scratch, float* loss,
                                float* d_flat, void* ws, size_t ws_bytes, void* stream) {
    if (!ctx || !out || !pixels || !loss || (kind != 0 && kind != 1) || n_rays < 1) return EONERF_E_ARG;
    if ((n_rays + 255) / 256 > LOSS_MAX_BLOCKS) {      // beyond the fused kernel's ticket sum: the two calls it replaces
        if (!d_out_scratch) return EONERF_E_ARG;
        const int rc = eonerf_train_loss(ctx, out, pixels, n_rays, kind, d_out_scratch, loss, stream);
        return rc ? rc : render_backward_impl(ctx, flat, rays, img_idx, n_rays, flags, d_out_scratch, nullptr, d_flat, ws, ws_bytes, stream);
    }
    const LossSpec ls{out, pixels, kind, loss};
    return render_backward_impl(ctx, flat, rays, img_idx, n_rays, flags, nullptr, &ls, d_flat, ws, ws_bytes, stream);
}

// diagnostics: copies the cycle sums of the last pipelined backward ([n_pipes * 7 roles][2 waves][8] u64) to the host; returns the
// number of u64 written (0 when EONERF_PIPE_STAMPS is off)
int eonerf_debug_pipe_stamps(eonerf_ctx* ctx, unsigned long long* host_out, int capacity) {
    if (!ctx || !ctx->pipe_stamps || !host_out) return 0;
    const int n = ctx->n_pipes * PIPE_STAGES * 128;
    if (capacity < n) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(host_out, ctx->pipe_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

int eonerf_device_status(eonerf_ctx* ctx, void* stream) {
    if (!ctx) return EONERF_E_ARG;
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, ctx->dev_status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (!err) return EONERF_OK;
    HIP_TRY(hipMemsetAsync(ctx->dev_status, 0, sizeof(int), (hipStream_t)stream));
    // only the presample guard: a training forward consumed samples eonerf_presample had drawn for OTHER ray contents (the caller refilled
    // the buffer in place); k_adam skipped that step's update like any flagged step.  A call-sequence error, not a device fault: the
    // pipelined path stays
    if ((err & ~0x100) == EO_STATUS_PRESAMPLE_STALE) return EONERF_E_STATE;      // (0x100: k_adam's echo of the sealed flag in the gradient message)
    // A hand-off timed out: the launch did not have the card's CUs to itself (a co-tenant, a partitioned or CU-masked GPU) or a stage
    // stalled.  The fault is reported (the caller decides: the launcher ends the job) and THIS context leaves the pipelined path: a
    // caller that carries on trains through the chain + GEMM backward instead of paying a 0.3-s timeout in every step.  Data-parallel
    // ranks all see the fault (flag in the gradient message) and all switch.  Backwards of forwards that ran BEFORE this call keep the
    // pipelined path (ws_pipe / PipeModeGuard: their workspace layout and mask slots are that path's).
    if (ctx->pipe && ctx->pipe_fallback) { ctx->pipe = false; ctx->need_repack = true; }
    return EONERF_E_DEVICE;
}

int eonerf_range_status(eonerf_ctx* ctx, void* stream) {
    if (!ctx) return EONERF_E_ARG;
    if (ctx->prec != EONERF_F16X3) return EONERF_OK;
    static_assert(RANGE_STICKY_WORD == RANGE_WORD + 1, "both words in one copy");
    int flag[2] = {0, 0};      // [0] operands of the calls since the last check, [1] the weights of the last re-pack (stays up until they change)
    HIP_TRY(hipMemcpyAsync(flag, ctx->dev_status + RANGE_WORD, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (!flag[0] && !flag[1]) return EONERF_OK;
    if (flag[0]) HIP_TRY(hipMemsetAsync(ctx->dev_status + RANGE_WORD, 0, sizeof(int), (hipStream_t)stream));
    return EONERF_E_RANGE;
}

int eonerf_render_status(eonerf_ctx* ctx, int n_rays, int flags, void* ws, size_t ws_bytes, void* stream) {
    (void)n_rays; (void)flags; (void)ws; (void)ws_bytes;
    return eonerf_device_status(ctx, stream);
}

size_t eonerf_grad_floats(const eonerf_ctx* ctx) { return ctx ? ctx->pl.total + 4 : 0; }
size_t eonerf_grad_early_floats(const eonerf_ctx* ctx) { return ctx ? ctx->pl.early : 0; }

int eonerf_set_exchange_event(eonerf_ctx* ctx, void* hip_event, int reserve_cus) {
    if (!ctx || reserve_cus < 0 || reserve_cus > ctx->n_cu / 2) return EONERF_E_ARG;
    ctx->exch_event = (hipEvent_t)hip_event;
    ctx->exch_cus = hip_event ? reserve_cus : 0;
    return EONERF_OK;
}

int eonerf_grad_seal(eonerf_ctx* ctx, float* d_flat, void* stream) {
    if (!ctx || !d_flat) return EONERF_E_ARG;
    return (int)eo_launch_grad_seal(d_flat + ctx->pl.total, ctx->dev_status, (hipStream_t)stream);
}

int eonerf_train_loss(eonerf_ctx* ctx, const float* out, const float* pixels, int n_rays, int kind, float* d_out, float* loss, void* stream) {
    if (!ctx || !out || !pixels || !d_out || !loss || n_rays < 1 || (kind != 0 && kind != 1)) return EONERF_E_ARG;
    return (int)eo_launch_loss(out, pixels, n_rays, kind, d_out, loss, ctx->loss_scratch, (hipStream_t)stream);
}

static int adam_common(eonerf_ctx* ctx, float* flat, float* d_flat, bool zero_grad, float* exp_avg, float* exp_avg_sq,
                       int step, float lr, float beta1, float beta2, float eps, float grad_scale, const float* fault_flag, void* stream) {
    if (!ctx || !flat || !d_flat || !exp_avg || !exp_avg_sq || step < 1) return EONERF_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(eo_launch_adam(flat, d_flat, zero_grad, exp_avg, exp_avg_sq, ctx->pl.total, step, lr, beta1, beta2, eps, grad_scale, ctx->dev_status, fault_flag, st));
    return eonerf_set_weights(ctx, flat, stream);
}
int eonerf_adam_step(eonerf_ctx* ctx, float* flat, const float* d_flat, float* exp_avg, float* exp_avg_sq,
                     int step, float lr, float beta1, float beta2, float eps, float grad_scale, const float* fault_flag, void* stream) {
    return adam_common(ctx, flat, const_cast<float*>(d_flat), false, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale, fault_flag, stream);
}
int eonerf_adam_step_zero_grad(eonerf_ctx* ctx, float* flat, float* d_flat, float* exp_avg, float* exp_avg_sq,
                               int step, float lr, float beta1, float beta2, float eps, float grad_scale, const float* fault_flag, void* stream) {
    return adam_common(ctx, flat, d_flat, true, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale, fault_flag, stream);
}

}  // extern "C"

