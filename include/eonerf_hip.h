/* eonerf_hip.h -- C ABI of libeonerf_hip.so: the MI355X (gfx950) implementation of the EO-NeRF per-ray hot path.
 *
 * The reference (rogermm14/eonerf_code) has NO native boundary of its own: its hot path is a Python API
 * (SURVEY.md 8b).  Each entry point below names the reference Python interface it sits under; the host-side
 * mirror of those interfaces lives in eonerf_code_amd/ (ctypes), INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers + sizes + a hipStream_t (passed as void*).
 *   - the caller (PyTorch) owns every buffer incl. the workspace; the library allocates device memory only in
 *     eonerf_create (packed weights, tables) and frees it in eonerf_destroy.
 *   - every call is asynchronous on `stream`; nothing synchronises the device.
 *   - return value: 0 = OK, < 0 = EONERF_E_* (bad argument / unsupported), > 0 = hipError_t.
 *   - one context per process/GPU, used from one host thread (the reference is single-threaded,
 *     train_eonerf.py:98-161).
 */
#ifndef EONERF_HIP_H
#define EONERF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EONERF_VERSION 502

enum { EONERF_OK = 0, EONERF_E_ARG = -1, EONERF_E_WORKSPACE = -2, EONERF_E_STATE = -3, EONERF_E_UNSUPPORTED = -4, EONERF_E_DEVICE = -5, EONERF_E_RANGE = -6 };

/* arithmetic of the MLP GEMMs */
enum { EONERF_FP32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32 FMA chains (parity mode, 1e-4 vs the reference) */
       EONERF_BF16 = 1,   /* v_mfma_f32_32x32x16_bf16, fp32 accumulate (throughput mode) */
       EONERF_F16X3 = 2 }; /* INFERENCE only (export renders): every operand as hi + lo fp16, three v_mfma_f32_32x32x16_f16 per product, fp32
                            * accumulate: fp32-level accuracy at 3/16 of the fp32 path's matrix time.  Training entry points (EONERF_F_TRAIN,
                            * *_train, *_backward) return EONERF_E_UNSUPPORTED */

/* flags of eonerf_render_forward */
enum { EONERF_F_SHADOWS = 1,     /* epoch_idx >= 2: shadow-ray pass + s = geo_shadow * transient_s (sat_rendering.py:269-276) */
       EONERF_F_EVAL = 2,        /* eval=True: radiometric row of the chunk's first ray (sat_rendering.py:288-291) */
       EONERF_F_TRAIN = 4,       /* keep what eonerf_render_backward needs in the workspace */
       EONERF_F_ONLY_DEPTH = 8,  /* only_depth=True branch (sat_rendering.py:227-249): out[:,3] only */
       EONERF_F_RGB_LOSS = 16 }; /* with F_TRAIN and without F_SHADOWS: the caller's loss depends on rgb/depth/albedo only, i.e.
                                  * d_out[:,11:13] (transient_s, beta) == 0 -- exactly the reference's epoch < 2 graph (s = 1,
                                  * F.mse_loss on rgb; train_eonerf.py:139-141, sat_rendering.py:269-272), where autograd never
                                  * visits the transient head.  Its activations are then not saved and its backward is skipped. */

typedef struct eonerf_ctx eonerf_ctx;

typedef struct {
    int n_images;        /* EONerfMLP(n_input_images), radiance_fields/eonerf.py:70-77 */
    int precision;       /* EONERF_FP32 | EONERF_BF16 | EONERF_F16X3 */
    int n_samples;       /* int(2/render_step_size) (sat_rendering.py:64): 2 .. 256 (128: run_JAX_RGB.sh:11); eonerf_set_n_samples changes it */
    int radiometric;     /* radiometric_normalization (opt.py:98-99 forces 1 for eo-nerf) */
} eonerf_config;

/* lifetime -- replaces EONerfMLP.__init__ / .to(device) (radiance_fields/eonerf.py:70-139, train_eonerf.py:60-61) */
int eonerf_create(eonerf_ctx** out, const eonerf_config* cfg);
int eonerf_destroy(eonerf_ctx* ctx);

/* Flat fp32 parameter buffer: every tensor of the reference state_dict lives at a fixed offset of one
 * contiguous device buffer (which is also the Adam / all-reduce unit).  Enumerate with index 0..n-1. */
int eonerf_param_tensors(const eonerf_ctx* ctx);
int eonerf_param_info(const eonerf_ctx* ctx, int index, const char** name, size_t* offset, int* rows, int* cols);
size_t eonerf_param_floats(const eonerf_ctx* ctx);

/* Re-pack the fp32 master weights into the LDS fragment-order streams the kernels read.  Call after every
 * optimizer step / load_state_dict (replaces nothing in the reference: nn.Linear reads its weights in place). */
int eonerf_set_weights(eonerf_ctx* ctx, const float* flat_params, void* stream);

/* workspace the caller must pass to the calls below for batches of up to n_points samples / n_rays rays */
size_t eonerf_field_workspace_bytes(const eonerf_ctx* ctx, int n_points);
size_t eonerf_render_workspace_bytes(const eonerf_ctx* ctx, int n_rays, int flags);

/* EONerfMLP.forward(x, sun_dirs, img_indices) -- radiance_fields/eonerf.py:154-170.
 * flat_params: the fp32 master buffer (embedding / ambient tables are read from it directly).
 * xyz[n,3], sun[n,3], img[n] (int64) -> sigma[n], albedo[n,3], ambient[n,3], ts[n], tb[n]  (row-major fp32). */
int eonerf_field_forward(eonerf_ctx* ctx, const float* flat_params, const float* xyz, const float* sun, const int64_t* img, int n,
                         float* sigma, float* albedo, float* ambient, float* ts, float* tb,
                         void* workspace, size_t workspace_bytes, void* stream);

/* EONerfMLP.query_density(x) -- radiance_fields/eonerf.py:141-145 (query_opacity = density * step on the host). */
int eonerf_query_density(eonerf_ctx* ctx, const float* flat_params, const float* xyz, int n, float* sigma,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same two calls as differentiable operators (the reference's EONerfMLP is an ordinary autograd module,
 * radiance_fields/eonerf.py:141-170): eonerf_field_forward_train also keeps, in `workspace`, what eonerf_field_backward needs
 * (activations, ReLU masks, positions); density_only != 0 is query_density (sun/img/albedo/ambient/ts/tb may be NULL).
 * eonerf_field_backward: upstream gradients g_* of the outputs (same shapes as the outputs, NULL = zero) -> parameter
 * gradients ACCUMULATED into d_flat_params and, if d_xyz != NULL, the input gradient d_xyz[n,3] (through the encoder
 * derivative 2^k cos(2^k x)).  Must follow a forward_train on the same (n, density_only, workspace). */
size_t eonerf_field_train_workspace_bytes(const eonerf_ctx* ctx, int n_points, int density_only);
int eonerf_field_forward_train(eonerf_ctx* ctx, const float* flat_params, const float* xyz, const float* sun, const int64_t* img, int n,
                               int density_only, float* sigma, float* albedo, float* ambient, float* ts, float* tb,
                               void* workspace, size_t workspace_bytes, void* stream);
int eonerf_field_backward(eonerf_ctx* ctx, const float* flat_params, const float* sun, int n, int density_only,
                          const float* g_sigma, const float* g_albedo, const float* g_ambient, const float* g_ts, const float* g_tb,
                          float* d_flat_params, float* d_xyz, void* workspace, size_t workspace_bytes, void* stream);

/* RPC camera model in rpcm's dict format (the "rpc" entry of the dataset JSON files, datasets/satellite.py:52-55). */
typedef struct {
    double col_num[20], col_den[20], row_num[20], row_den[20];
    double row_offset, col_offset, lat_offset, lon_offset, alt_offset;
    double row_scale, col_scale, lat_scale, lon_scale, alt_scale;
} eonerf_rpc;

/* datasets/satellite.py get_rays (:65-121, utm branch) + get_sun_dirs (:486-500) + normalize_rays (:124-139) for one image:
 * pixels (cols[n], rows[n] fp64 device arrays, or NULL/NULL for the full width x height grid in row-major order) ->
 * raw8[n,8] fp32 (origin UTM/alt, unit dir, near=0, far; the payload of the reference's ray cache files, may be NULL) and
 * rays[n,11] fp32 normalised rays incl. the sun direction (may be NULL).  utm_zone/south select "+proj=utm +zone=.. [+south]"
 * (sat_utils.py:99-116); sun_elevation_deg/sun_azimuth_deg are the JSON's values (the 90-elevation flip of :457 happens
 * inside); offset/scale are scene.loc_utm's X/Y/Z values.  RPC localisation (rpcm) and the UTM projection (PROJ) are
 * re-implemented from their published algorithms in fp64.  geo (may be NULL): the fp64 intermediates [n,8] = lon, lat (degrees),
 * UTM east, north (metres) of the localised point at max_alt, then at min_alt -- what rpc.localization and
 * utm_from_latlon return BEFORE the fp32 cast of datasets/satellite.py:119-120 (test / diagnostic output). */
int eonerf_generate_rays(const eonerf_rpc* rpc, const double* cols, const double* rows, long n, int width,
                         double min_alt, double max_alt, int utm_zone, int south,
                         double sun_elevation_deg, double sun_azimuth_deg, const float offset[3], const float scale[3],
                         float* raw8, float* rays, double* geo, void* stream);

/* n_samples = int(2 / render_step_size) of the calls that follow (sat_rendering.py:64, opt.py:54): any value from 2 to 256 (version 502; 64 / 128 / 256 before) -- the per-ray kernels are
 * instantiated for 1, 2 and 4 samples per lane of the ray's wavefront; anything else: EONERF_E_UNSUPPORTED.  Every size below that is
 * written with 128 / 127 (zsteps, jitter arrays, samples per ray, workspace sizes) follows it.  A backward call must run under the
 * n_samples of its forward (the Python layer restores it). */
int eonerf_set_n_samples(eonerf_ctx* ctx, int n_samples);

/* Key of the in-kernel jitter stream (torch.manual_seed's role for perturb_z_vals' rand_like, sat_rendering.py:52;
 * data-parallel ranks use different seeds).  Every call that draws noise advances the stream. */
int eonerf_set_noise_seed(eonerf_ctx* ctx, uint64_t seed);

/* sat_rendering.satnerf_sampling (sat_rendering.py:56-84) + count_number_of_pts_per_nerfacc_ray (:10-16):
 * rays[R,11] (origin, dir and near columns are used), u[R,n_samples] jitter -> flattened, cube-filtered samples
 * ray_indices[n] (int64), t_starts[n], t_ends[n] (capacity R*(n_samples-1) each), pts_per_ray[R] (fp32) and *n_dev = n.
 * perturb = 0: the z values stay on the uniform grid (perturb=False, :70-71 skipped; u is ignored).  perturb != 0 with
 * u == NULL: the jitter is drawn inside the kernel (Philox4x32-10, eonerf_set_noise_seed). */
int eonerf_sample_rays(eonerf_ctx* ctx, const float* rays, const float* zsteps, const float* u, int perturb, int n_rays,
                       int64_t* ray_indices, float* t_starts, float* t_ends, float* pts_per_ray, int* n_dev,
                       void* workspace, size_t workspace_bytes, void* stream);

/* EONerfMLP.rendering / render_depth on caller-provided flattened samples (radiance_fields/eonerf.py:172-248):
 * ray_indices must be sorted (as satnerf_sampling returns them).  depth_only != 0 -> only depth[R] is written.
 * Outputs per ray: albedo[R,3], depth[R], beta[R] (incl. +beta_min), transient_s[R], ambient[R,3] (before the x0.2 of
 * render_image), entropy[R] (ones).  Inference entry point (no saved state for backward). */
int eonerf_rendering(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx,
                     const float* t_starts, const float* t_ends, const int64_t* ray_indices, int n, int n_rays, int depth_only,
                     float* albedo, float* depth, float* beta, float* transient_s, float* ambient, float* entropy,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same two entry points under autograd (the reference's methods are ordinary autograd code, radiance_fields/eonerf.py:172-248):
 *   eonerf_rendering_train: as eonerf_rendering, with the chain kernel in training mode; `workspace` (eonerf_render_workspace_bytes with
 *     EONERF_F_TRAIN [| EONERF_F_ONLY_DEPTH]) keeps the saved activations and compositing inputs and must stay untouched until
 *   eonerf_rendering_backward: gradients of the per-ray outputs ([n_rays, c] row-major, NULL = zero; entropy is a constant) -> gradient of
 *     every parameter, ACCUMULATED into d_flat_params.  t_starts / t_ends themselves carry no gradient (the reference's samplers run
 *     under no_grad, sat_rendering.py:56). */
int eonerf_rendering_train(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx,
                           const float* t_starts, const float* t_ends, const int64_t* ray_indices, int n, int n_rays, int depth_only,
                           float* albedo, float* depth, float* beta, float* transient_s, float* ambient, float* entropy,
                           void* workspace, size_t workspace_bytes, void* stream);
int eonerf_rendering_backward(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx, int n_rays, int depth_only,
                              const float* g_albedo, const float* g_depth, const float* g_beta, const float* g_transient_s, const float* g_ambient,
                              float* d_flat_params, void* workspace, size_t workspace_bytes, void* stream);

/* One chunk of sat_rendering.render_image (sat_rendering.py:252-312) = satnerf_sampling + EONerfMLP.rendering +
 * compute_geometric_shadows + irradiance/radiometric model + output packing.
 *   rays[R,11] fp32 (o3 d3 near far sun3), img_idx[R] int64, zsteps[n_samples] = linspace(0,1,n_samples) (n_samples: 128 unless
 *   eonerf_set_n_samples said otherwise),
 *   u_cam/u_sun[R,n_samples] jitter in [0,1) (the reference draws rand_like inside perturb_z_vals, :52),
 *   u_retry (may be NULL): noise of the ":260-262 resample if some ray is empty" branch.
 *   u_cam == NULL (then u_retry and u_sun must be NULL too): production mode, no noise buffers -- the sampler kernels draw the
 *   jitter themselves (Philox4x32-10 keyed by eonerf_set_noise_seed, counter = (ray, sample lane, draw, call number); the
 *   reference's torch.rand_like is the same generator family, :52), and the resample branch is always armed.
 *   out[R,21] = rgb3 depth1 albedo3 ambient3 geo1 ts1 beta1 entropy1 pts1 sc_pts1 opacity2 shadowless3 (:311-312)
 *   n_samples_dev: device int, number of camera samples (render_image's second return value).
 *   Size limit with EONERF_F_TRAIN: 66,050 rays per call in bf16 mode, 33,024 in fp32 mode (a 256-row block of the saved-activation
 *   slabs is addressed with 32-bit byte offsets); EONERF_E_UNSUPPORTED beyond -- chunk the batch, as render_image's `chunk` does. */
int eonerf_render_forward(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx,
                          const float* zsteps, const float* u_cam, const float* u_retry, const float* u_sun,
                          int n_rays, int flags, float* out, int* n_samples_dev,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The camera pass's sampler of the NEXT eonerf_render_forward, launched ahead of it (EONERF_F_TRAIN, production noise only): it depends on
 * the rays and the noise seed, not on the weights, so a data-parallel trainer enqueues it on the compute stream while the gradient
 * all-reduce of the step before runs on its own stream (the reference has no counterpart: single process, train_eonerf.py:98-161; SURVEY.md
 * 8e).  The forward that follows with the SAME rays / img_idx / zsteps / n_samples_dev pointers, n_rays, flags and workspace skips its sampler launches and
 * draws the shadow pass under the same Philox call number: results are bit-identical to a forward that samples itself.  Any other call that
 * writes the workspace drops the record (the next forward samples again); a backward on that workspace in between returns EONERF_E_STATE.
 * The workspace must be free: the backward that last used it has been enqueued on `stream`. */
int eonerf_presample(eonerf_ctx* ctx, const float* rays, const int64_t* img_idx, const float* zsteps, int n_rays, int flags,
                     int* n_samples_dev, void* workspace, size_t workspace_bytes, void* stream);
/* The record is matched to its forward by pointer identity, so the CONTENTS of rays / img_idx must not change between the two calls.  Two
 * guards (version 502): a caller that knows it refilled a buffer calls eonerf_presample_cancel (the next forward samples itself; the
 * Python trainer does this when a hinted tensor's version counter moved); and the library checks on the device: the presampling sampler
 * sums a digest of the ray words it reads (origin, direction, near, image index), the backward of the forward that consumed the record
 * sums it again from the same buffers, and a mismatch raises bit 9 of the context's sticky status word -- eonerf_adam_step skips that
 * step's update (its samples belonged to other rays) and the next eonerf_device_status returns EONERF_E_STATE (not EONERF_E_DEVICE: the
 * pipelined path is not left). */
int eonerf_presample_cancel(eonerf_ctx* ctx);

/* Autograd of the call above (loss.backward(), train_eonerf.py:160): d_out[R,21] -> gradient of every parameter,
 * ACCUMULATED into d_flat_params (same layout as the flat parameter buffer).  `workspace` must be the one a
 * render_forward with EONERF_F_TRAIN and the same (rays, img_idx, n_rays, flags) filled; nothing else is remembered
 * between the two calls, so several forward chunks may be outstanding, each with its own workspace. */
int eonerf_render_backward(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx,
                           int n_rays, int flags, const float* d_out, float* d_flat_params,
                           void* workspace, size_t workspace_bytes, void* stream);

/* eonerf_train_loss + eonerf_render_backward in one call (train_eonerf.py:139-143 and :160 back to back, what a training loop does): the
 * loss of `kind` on the forward's packed outputs out[R,21] against pixels[R,3] -> *loss (device scalar, bit-identical to eonerf_train_loss's)
 * and, without d out[R,21] ever being written, the gradients of every parameter ACCUMULATED into d_flat_params -- the loss gradient is
 * formed inside the backward's first kernel (one launch fewer per step).  d_out_scratch[R,21] is only used beyond 65,536 rays per call
 * (there the two calls run one after the other) and may be NULL below. */
int eonerf_render_backward_loss(eonerf_ctx* ctx, const float* flat_params, const float* rays, const int64_t* img_idx,
                                int n_rays, int flags, const float* out, const float* pixels, int kind, float* d_out_scratch, float* loss,
                                float* d_flat_params, void* workspace, size_t workspace_bytes, void* stream);

/* Device-side health.  In bf16 mode the trunk / heads backward are persistent, layer-pipelined kernels whose workgroups hand
 * tiles to each other inside the launch; every wait in them is bounded by a wall-clock watchdog, so a launch always drains, and a
 * wait that expired is recorded in a STICKY status word owned by the context: no later launch clears it, eonerf_adam_step refuses
 * to apply gradients while it is set (see there), and only eonerf_device_status reads and clears it.
 *   eonerf_device_status: SYNCHRONISES `stream`, returns EONERF_E_DEVICE if a fault was recorded since the last call (0 otherwise)
 *   and clears the word.  Call it where the host synchronises anyway (the reference's loop reads the loss every 1000 steps,
 *   train_eonerf.py:173-178) -- on EVERY data-parallel rank.
 *     After a reported fault the context leaves the pipelined path for the rest of its life (its hand-offs need the card's CUs to
 *     themselves: a co-tenant, a partitioned or CU-masked GPU): later steps run the chain + GEMM backward (EONERF_PIPE=0's path) instead
 *     of timing out again; EONERF_PIPE_FALLBACK=0 keeps the pipeline.  eonerf_create makes the same choice up front when the kernel
 *     cannot be resident at all or the process runs under a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK), unless EONERF_PIPE=1 insists.
 *   eonerf_render_status: the same check, kept with the workspace arguments of version 2 of this ABI (they are ignored). */
int eonerf_device_status(eonerf_ctx* ctx, void* stream);
/* EONERF_F16X3 contexts: the RANGE of the split precision.  An operand is carried as hi + lo fp16: exact to ~2^-21 relative for
 * 2^-14 <= |v| <= 65504; below 2^-14 the lo half is an fp16 subnormal (ABSOLUTE accuracy 2^-25, still far below what a weight or an
 * activation of that size contributes to a sum next to O(1) terms); above 65504 hi rounds to infinity and the value is lost (the reference's
 * fp32 arithmetic has no such limit).  Every fp16 x 3 kernel therefore flags an activation, an embedding value or a position outside
 * +-65504 (or not finite) in a context-owned word, and every re-pack (eonerf_set_weights) flags a weight matrix whose largest |w| is
 * not finite, above 64 (absolute operand errors amplified beyond "fp32-level") or below 2^-9 (the whole matrix in the subnormal-lo regime).  eonerf_range_status SYNCHRONISES `stream`, returns EONERF_E_RANGE if the flag was
 * raised since the last call (0 otherwise, always 0 for the other precisions) and clears it; the Python layer calls it where an export
 * render synchronises anyway and repeats the call on an EONERF_FP32 context (sat_rendering.render_image, EONerfMLP eval-mode queries). */
int eonerf_range_status(eonerf_ctx* ctx, void* stream);
int eonerf_render_status(eonerf_ctx* ctx, int n_rays, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* The data-parallel gradient MESSAGE (one flat fp32 all-reduce per step) is eonerf_grad_floats() long:
 * the eonerf_param_floats() gradients followed by 4 control floats, [0] = fault flag.  eonerf_grad_seal writes the flag
 * (1.0 if this context's status word is set, else 0.0) behind the last backward of a step, so that the ONE sum all-reduce also
 * tells every rank that some rank's gradients are invalid; every rank then skips the update (eonerf_adam_step) and raises at its
 * next eonerf_device_status.  d_flat_params must hold eonerf_grad_floats() floats for this call. */
size_t eonerf_grad_floats(const eonerf_ctx* ctx);
int eonerf_grad_seal(eonerf_ctx* ctx, float* d_flat_params, void* stream);
/* Two-bucket exchange (version 502).  The first eonerf_grad_early_floats() floats of the message are the trunk layers the camera pass'
 * layer-pipelined launch completes (layers 1-4, 6, 7, weights and biases: 58 % of the message); every later gradient kernel of the
 * backward (the GEMM launch, the tail) writes only behind them.  eonerf_set_exchange_event(ctx, hipEvent_t, reserve_cus) arms an event
 * that every eonerf_render_backward[_loss] records on its stream at that point (chain + GEMM path: at the end of the call), so that a
 * trainer can start the all-reduce of [0, early) on its communication stream while the rest of the backward runs, and the one of
 * [early, eonerf_grad_floats()) behind eonerf_grad_seal ("issue ... right after the last dW").  reserve_cus CUs are left
 * out of the grids of the gradient kernels behind that point: they fill every CU they are given for ~0.5 ms, and the collective's
 * kernel needs one to run beside them.  NULL disarms.  The event must outlive the calls that record it. */
size_t eonerf_grad_early_floats(const eonerf_ctx* ctx);
int eonerf_set_exchange_event(eonerf_ctx* ctx, void* hip_event, int reserve_cus);

/* Training loss on the packed outputs and its gradient (train_eonerf.py:139-143): kind 0 = F.mse_loss(rgb, pixels),
 * kind 1 = metrics.uncertainty_aware_loss(pixels, rgb, beta) (metrics.py:17-22, including the constant 3/2 of its beta term).
 * Writes d_out[R,21] (zero except the rgb/beta columns) and the scalar *loss (device). */
int eonerf_train_loss(eonerf_ctx* ctx, const float* out, const float* pixels, int n_rays, int kind, float* d_out, float* loss, void* stream);

/* torch.optim.Adam step on the flat buffers (train_eonerf.py:63,161): lr, betas (0.9,0.999), eps 1e-8, no weight decay; ONE step
 * count for every parameter (the reference's cat + slice graph hands the transient / ambient heads defined ZERO gradients while
 * epoch_idx < 2 -- sat_rendering.py:294,311-312,322 -- so torch.optim.Adam steps them from step 1 with a zero update).
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
 * fault_flag (device pointer, may be NULL): the update is SKIPPED -- parameters and moments untouched -- when *fault_flag != 0
 * (the reduced flag of eonerf_grad_seal: some rank's gradients are invalid) or when this context's sticky status word is set;
 * a skip through fault_flag also sets the status word, so the next eonerf_device_status on this rank reports it. */
int eonerf_adam_step(eonerf_ctx* ctx, float* flat_params, const float* d_flat_params, float* exp_avg, float* exp_avg_sq,
                     int step, float lr, float beta1, float beta2, float eps, float grad_scale, const float* fault_flag, void* stream);
/* The same, and the gradient message is CONSUMED: every float of d_flat_params is zero afterwards -- optimizer.step() fused with the
 * optimizer.zero_grad() of the next iteration (train_eonerf.py:158-161), also when the update is skipped (invalid gradients must not
 * leak into the next accumulation).  The 4 control floats behind the gradients are left to eonerf_grad_seal. */
int eonerf_adam_step_zero_grad(eonerf_ctx* ctx, float* flat_params, float* d_flat_params, float* exp_avg, float* exp_avg_sq,
                               int step, float lr, float beta1, float beta2, float eps, float grad_scale, const float* fault_flag, void* stream);

/* Measurement hooks (no reference counterpart): with profiling enabled every launch of the MFMA kernels is bracketed by hipEvents on
 * the caller's stream, ONE scope per kernel launch (so a scope's time is that kernel's time).  In bf16 mode the backward of a pass is
 * heads chain -> trunk pipeline -> [input-gradient tail], then one weight-gradient GEMM for the jobs the pipeline
 * leaves; in fp32 mode (or EONERF_PIPE=0) the two chain scopes cover the whole dX chain and the GEMM every weight gradient.
 * eonerf_profile_read synchronises on the recorded events and returns the summed duration and launch count;
 * eonerf_profile_name gives the scope's name (NULL beyond the last). */
enum { EONERF_PROF_FWD_CHAIN_CAMERA = 0, EONERF_PROF_BWD_CHAIN_CAMERA = 1, EONERF_PROF_WGRAD = 2, EONERF_PROF_FWD_CHAIN_SUN = 3,
       EONERF_PROF_BWD_CHAIN_SUN = 4, EONERF_PROF_BWD_PIPE_CAMERA = 5, EONERF_PROF_BWD_PIPE_SUN = 6, EONERF_PROF_IG_TAIL_SUN = 7,
       EONERF_PROF_KERNELS = 8 };
int eonerf_profile_enable(eonerf_ctx* ctx, int max_launches_per_kernel);
int eonerf_profile_read(eonerf_ctx* ctx, int kernel, float* total_ms, int* launches);
const char* eonerf_profile_name(int kernel);
/* Clock probe (no reference counterpart; bench.py's attribution of a slow step): one launch of a FIXED dense bf16 MFMA loop on every CU
 * (8192 v_mfma_f32_32x32x16_bf16 per wave, 4 waves per CU; ~0.13 ms at 2 GHz), read in-kernel against the shader-clock counter (s_memtime)
 * and the constant 100-MHz counter (s_memrealtime).  out4 (DEVICE, 4 floats): [0] shader cycles, [1] 100-MHz ticks (x 10 ns = the probe's
 * duration: fixed work, so it is inversely proportional to the clock the chip held), [2] cycles / ticks x 100 = shader clock in MHz. */
int eonerf_clock_probe(eonerf_ctx* ctx, float* out4, void* stream);

const char* eonerf_strerror(int code);
int eonerf_version(void);

#ifdef __cplusplus
}
#endif
#endif
