// Argument blocks of the per-ray kernels (eonerf_rays.hip, eonerf_rays_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// per-ray record produced by compositing (floats)
constexpr int RR_DEPTH = 0, RR_ALB = 1, RR_TS = 4, RR_TB = 5, RR_WSUM = 6, RR_AMB = 7, RR_GEO = 10, RAY_REC = 12;

struct AmbientW { const float *w1, *b1, *w2, *b2; };   // ambient_mlp: [128][27], [128], [3][128], [3]  (fp32)

// Content digest of what the camera sampler consumes of a ray -- words 0..6 of its table row (origin, direction, near) and its image
// index -- summed over the rays (order independent).  eonerf_presample leaves it beside its samples; the backward of the forward that
// consumed them recomputes it from the buffers as they are THEN and raises the context's status word on a mismatch (a caller refilled
// the ray buffer in place between the two calls: the samples belong to the old rays).
__host__ __device__ inline unsigned long long ray_word_digest(uint32_t word, uint32_t index) {
    uint32_t h = word ^ (0x9E3779B9u * (index + 1u));
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (unsigned long long)h * 0x9E3779B97F4A7C15ull + (unsigned long long)word;
}
constexpr int EO_STATUS_PRESAMPLE_STALE = 1 << 9;      // bit of the context's status word (eonerf_device_status -> EONERF_E_STATE)

struct SampleArgs {
    const float* rays;        // [R][11] fp32: o3 d3 near far sun3  (datasets/satellite.py:23-26)
    const int64_t* img_idx;   // [R] or nullptr
    int n_samples;            // int(2 / render_step_size): 2 .. 256 (sat_rendering.py:64); a ray has n_samples - 1 intervals
    const float* zsteps;      // [n_samples] = torch.linspace(0,1,n_samples)
    const float* u;           // [R][n_samples] jitter of this pass, or nullptr: drawn in the kernel (Philox4x32-10, seed/call below)
    const float* u_retry;     // [R][n_samples] noise of the "some ray is empty -> resample" branch (nullptr with retry: Philox)
    int retry;                // 1: the resample branch exists (camera pass of render_image, sat_rendering.py:260-262)
    int perturb;              // 0: perturb=False, z_vals stay on the uniform grid (sat_rendering.py:70-71 skipped)
    uint64_t seed; uint32_t call;   // Philox key and the per-call word of its counter
    const float* depth;       // sun pass: rendered depth per ray
    int depth_stride;
    int n_rays;
    int sun_pass;             // 1: origin = o + depth*d, dir = -sun, near = 0
    int patch_last;           // 1: last interval of each ray ends at 1e10 (camera pass)
    int *cnt_first, *cnt_retry, *counts, *offsets;   // [R], [R], [R], [R+1]
    int* flags;               // bit0: retry taken
    int* n_pts;
    int* n_pts_copy;          // optional second destination of the sample count (the caller's device scalar), or nullptr
    float *px, *py, *pz, *tmid, *delta;   // [p_pad] compact outputs
    int* simg;
    // optional flattened outputs of satnerf_sampling (sat_rendering.py:82-84): ray_indices, t_starts, t_ends
    int64_t* o_ray; float *o_ts, *o_te;
    unsigned long long* digest;   // eonerf_presample: += the digest of every ray the emit kernel reads (zeroed by the caller), or nullptr
};

struct ShadeArgs {
    const float* ray_rec;
    const int64_t* img_idx;
    const float* radiometric;   // [n_img][9] or nullptr
    const int *pts_first, *sc_counts;
    int n_rays, use_shadow, eval;
    float* out;                 // [R][21]
};

struct CompositeArgs {
    int n_samples;            // as SampleArgs::n_samples
    const float* rays;
    const int *offsets, *counts;
    const float *sigma, *delta, *tmid, *albedo, *ts, *tb;
    int p_pad, n_rays;
    int shadow_only, depth_only;
    AmbientW amb;
    float* ray_out;           // [R][RAY_REC]
    float* amb_save;          // training: [R][160] = sun encoding (27, padded to 32) + hidden activations (128) of the ambient head
    // render_image's last compositing launch of a chunk also shades its ray and packs the 21 output columns (the ray record is
    // complete then: one launch fewer); do_shade = 0: EONerfMLP.rendering / render_depth, and the camera pass when a sun pass follows
    int do_shade;
    ShadeArgs shade;
    // the camera pass of a chunk with the shadow pass on also COUNTS the samples of the ray's shadow ray (its origin needs only this
    // ray's depth): the sun sampler then starts at its scan kernel.  count_sun = 0: no sun pass follows
    int count_sun;
    SampleArgs sun;
};

struct ShadeBwdArgs {
    const float *ray_rec, *d_out;
    const int64_t* img_idx;
    const float* radiometric;
    float* d_radiometric;       // inside the flat gradient buffer, or nullptr
    float* g_ray;               // [R][RAY_REC] gradient of the per-ray record
    int n_rays, use_shadow, eval;
    int lds_images;             // > 0: accumulate d_radiometric[n_img][6] in LDS first (n_img <= 2048)
    float* d_rad_rays;          // deterministic mode: [R][6] per-ray contributions instead of atomics (summed in ray order afterwards)
    // the first kernel of a backward call also zeroes what the call's later launches need zeroed (bottleneck factors, GEMM work queue,
    // the pipelined launches' sync blocks: one contiguous region of the workspace) -- a memset launch fewer; nullptr: nothing
    uint32_t* zero_base; size_t zero_bytes;      // 16-byte aligned, a multiple of 16 bytes
    // loss_kind >= 0: d_out is NOT read -- the training loss of eo_launch_loss (kind 0 MSE, 1 uncertainty-aware) is evaluated here on the
    // forward's packed outputs `loss_out` [R][21] against `loss_gt` [R][3], its gradient feeds the shading backward directly and the
    // scalar goes to *loss (same fixed-order ticket sum, scratch as in eo_launch_loss): one launch for train_eonerf.py:139-143 + the head
    // of :160.  Needs (n_rays + 255) / 256 <= LOSS_MAX_BLOCKS
    int loss_kind; const float *loss_out, *loss_gt; float *loss, *loss_scratch;
    // the forward of this backward consumed an eonerf_presample record: chk_sum += the digest of the rays as they are now (SampleArgs::digest)
    const float* chk_rays; unsigned long long* chk_sum;
};

struct CompositeBwdArgs {
    int n_samples;            // as SampleArgs::n_samples
    const float* rays;
    const int *offsets, *counts;
    const float *sigma, *delta, *tmid, *albedo, *ts, *tb;
    int p_pad, n_rays;
    const float* ray_rec;
    float* g_ray;
    float *g_sigma, *g_albedo, *g_ts, *g_tb;   // per-sample outputs
    const float* g_pos;                        // sun pass: [3][p_pad] d sigma / d position
    int depth_only;                            // camera compositing of a density-only pass (render_depth): no head terms
    // camera compositing backward with the shadow pass on: d depth first collects <d position, viewdir> over the ray's shadow samples
    // (origin = o + depth * d, sat_rendering.py:90) -- formerly a launch of its own (k_sun_depth_grad)
    const int *sun_offsets, *sun_counts;
    const float* sun_g_pos;                    // [3][p_pad] or nullptr
    // camera compositing backward: the digest the presampling sampler left against the one k_shade_bwd has just summed (nullptr: no check)
    const unsigned long long *chk_a, *chk_b; int* chk_status;
};

struct AmbientBwdArgs {
    AmbientW w;
    const float *rays, *ray_rec, *g_ray, *amb_save;
    int n_rays;
    float *d_w1, *d_b1, *d_w2, *d_b2;
};

// The bottleneck layer has an identity activation: bott = W_bott X8 + b_bott feeds the first layers of the albedo and transient heads.
// With the factors  M_a = dA1^T X8  (and M_t = dT1^T X8) and the bias gradients db_A1 = sum_p dA1 (db_T1) -- ONE weight-gradient job
// against X8 -- three weight gradients follow without the d bottleneck tensor and without the bottleneck OUTPUT ever being saved:
//     dW_bott += W_A1^T M_a (+ W_T1^T M_t)             db_bott += W_A1^T db_A1 (+ W_T1^T db_T1)          (d bottleneck itself is never formed: eonerf_pack.h)
//     dW_A1   += M_a W_bott^T + db_A1 (x) b_bott       (since sum_p dA1 bott^T = sum_p dA1 (W_bott X8 + b_bott)^T)     db_A1 -> d_flat
//     dW_T1[:, :256] += M_t W_bott^T + db_T1 (x) b_bott                                                               db_T1 -> d_flat
struct BottWgradArgs {
    const float *w_a1, *m_a;             // [128][256] weights, [128][256] dA1^T X8
    const float *w_t1, *m_t;             // transient head's first layer ([128][260] weights) or nullptr, [128][256] dT1^T X8
    const float* db_at;                  // [256] scratch: db_A1 | db_T1 of THIS backward call (not yet in the flat gradient buffer)
    const float *w_bott_t, *b_bott;      // W_bott TRANSPOSED ([in j][out i], the copy k_fold leaves beside the folded weights at every re-pack), [256]
    float *d_w, *d_b;                    // bottleneck layer: [256][256], [256] inside the flat gradient buffer
    float *d_w_a1, *d_b_a1;              // [128][256], [128]
    float *d_w_t1, *d_b_t1;              // [128][260] (columns 0..255 written), [128]; nullptr without the transient head
};

struct EmbGradArgs {
    int n_samples;           // as SampleArgs::n_samples
    const int *offsets, *counts;
    const int64_t* img_idx;
    const float* g_emb;      // [p_pad][4]
    float* d_emb;            // [n_img][4] inside the flat gradient buffer
    int n_rays;
    int lds_images;          // > 0: accumulate per block in LDS first
    float* d_emb_rays;       // deterministic mode: [R][4] per-ray sums instead of atomics (summed in ray order afterwards)
};

struct PackedArgs {        // flattened samples handed in by the caller (EONerfMLP.rendering / render_depth)
    const float* rays; const int64_t* img_idx;
    const float *t_starts, *t_ends; const int64_t* ray_indices;
    int n, n_rays;
    int *counts, *offsets, *n_pts;
    float *px, *py, *pz, *tmid, *delta; int* simg;
};
struct RenderingOutArgs { const float* ray_rec; int n_rays; float *albedo, *depth, *beta, *ts, *ambient, *entropy; };
// gradients of EONerfMLP.rendering's per-ray outputs (any may be null = zero) -> gradient of the ray record
struct RenderingOutBwdArgs { const float* ray_rec; int n_rays; const float *g_albedo, *g_depth, *g_beta, *g_ts, *g_ambient; float* g_ray; };
// bottleneck-factor products | embedding gradient | ambient-head backward in one launch (any of the three may be null)
constexpr int ENC_PART_F = 2 * 256 * 64 + 256;
// sum over the workgroups' partials -> accumulated into the flat gradient buffer (a role of k_step_tail)
struct EncPartReduceArgs {
    const float* part; int n_wg;
    float *dw0, *db0;         // layer 0: [256][63] and [256] inside the flat gradient buffer
    float* dw5s;              // layer 5: column 256 of [256][319] (its 63 skip columns)
    const int* col_map;       // [64] encoding slot -> reference column, -1 = padding slot
};
hipError_t eo_launch_step_tail(const BottWgradArgs* bott, const EmbGradArgs* emb, const AmbientBwdArgs* amb, hipStream_t st, const EncPartReduceArgs* enc = nullptr);
hipError_t eo_launch_rendering_out_bwd(const RenderingOutBwdArgs& a, hipStream_t st);

hipError_t eo_launch_sampler(const SampleArgs& a, hipStream_t st, bool counted = false);      // counted: cnt_first is filled already (CompositeArgs::count_sun)
hipError_t eo_launch_from_packed(const PackedArgs& a, hipStream_t st);
hipError_t eo_launch_rendering_out(const RenderingOutArgs& a, hipStream_t st);
hipError_t eo_launch_int_to_float(const int* src, int n, float* dst, hipStream_t st);
hipError_t eo_launch_shade_bwd(const ShadeBwdArgs& a, hipStream_t st);
hipError_t eo_launch_sun_composite_bwd(const CompositeBwdArgs& a, hipStream_t st);
hipError_t eo_launch_cam_composite_bwd(const CompositeBwdArgs& a, hipStream_t st);
hipError_t eo_launch_ambient_bwd(const AmbientBwdArgs& a, hipStream_t st, bool deterministic = false);
// out[idx[r] * stride + w] += contrib[r * width + w], summed over the rays in ray order by ONE thread per (table row, w)
hipError_t eo_launch_table_reduce(const float* contrib, const int64_t* idx, int n_rays, int width, int stride, int n_rows, int eval_first,
                                  float* out, hipStream_t st);
hipError_t eo_launch_bott_wgrad(const BottWgradArgs& a, hipStream_t st);
hipError_t eo_launch_field_grads_to_soa(const float* g_sigma, const float* g_albedo, const float* g_ts, const float* g_tb, int n, int p_pad,
                                       float* o_sigma, float* o_albedo, float* o_ts, float* o_tb, hipStream_t st);
hipError_t eo_launch_emb_grad_points(const float* g_emb, const int* simg, int n, float* d_emb, hipStream_t st);
hipError_t eo_launch_ambient_points_bwd(const AmbientW& w, const float* sun, const float* g_amb, int n,
                                        float* d_w1, float* d_b1, float* d_w2, float* d_b2, hipStream_t st);
hipError_t eo_launch_emb_grad(const EmbGradArgs& a, hipStream_t st);
constexpr int LOSS_MAX_BLOCKS = 256;      // k_loss: grid-stride over the rays, one partial sum per block in the context's scratch
hipError_t eo_launch_loss(const float* out, const float* gt, int n, int kind, float* d_out, float* loss, float* scratch, hipStream_t st);
// status: the context's sticky device status word; fault_flag: reduced fault flag of the gradient message or nullptr (see k_adam)
hipError_t eo_launch_adam(float* p, float* g, bool zero_grad, float* m, float* v, size_t n, int step, float lr, float b1, float b2, float eps,
                          float gscale, int* status, const float* fault_flag, hipStream_t st);
hipError_t eo_launch_grad_seal(float* tail, const int* status, hipStream_t st);
hipError_t eo_launch_composite_fwd(const CompositeArgs& a, hipStream_t st);
hipError_t eo_launch_shade_fwd(const ShadeArgs& a, hipStream_t st);
hipError_t eo_launch_points_to_soa(const float* xyz, const int64_t* img, int n, int p_pad, float* px, float* py, float* pz,
                                   int* simg, int* n_pts, hipStream_t st);
hipError_t eo_launch_ambient_points(const AmbientW& w, const float* sun, int n, float* out, hipStream_t st);
hipError_t eo_launch_soa3_to_aos(const float* soa, int p_pad, int n, float* aos, hipStream_t st);
