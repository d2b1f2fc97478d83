// Host<->kernel argument blocks and layout constants shared by the launchers (eonerf_api.cpp) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eonerf_common.h"
#include "eonerf_rays.h"

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of the kernel ON ONE DEVICE (of the code object loaded there): a process
// that drives several GPUs has to set it once per device, not once per process.
struct EoAttrOnce {
    bool done[64] = {};
    template <class F> hipError_t ensure(F&& set) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set();
        if (done[dev]) return hipSuccess;
        const hipError_t e = set();
        if (e == hipSuccess) done[dev] = true;
        return e;
    }
};

// ---- rows of the saved-activation slab [ACT_ROWS][p_pad] (feature-major, see eonerf_mlp_fwd.hip) ----
constexpr int ACT_ROW_ENC = 0;        // 64 encoding slots (slot order, see enc_col_of_hq)
constexpr int ACT_ROW_X1 = 64;        // X1..X8 : outputs of trunk layers 0..7, 256 rows each
constexpr int ACT_ROW_HEADS = 64 + 8 * 256;         // 2112 (the bottleneck OUTPUT has no rows: the layer is folded into the heads, eonerf_pack.h)
constexpr int ACT_ROW_A1 = ACT_ROW_HEADS;           // 2112, albedo hidden (128)
constexpr int ACT_ROW_T1 = ACT_ROW_A1 + 128;        // 2240, transient hidden T1..T4 (4 x 128)
constexpr int ACT_ROW_EMB = ACT_ROW_T1 + 512;       // 2752, transient embedding rows (8 used, 4 real)
constexpr int ACT_ROWS_FULL = ACT_ROW_EMB + 32;     // 2784
constexpr int ACT_ROWS_DENSITY = ACT_ROW_HEADS;     // 2112 (trunk only)

// ---- rows of the saved-gradient slab [GRD_ROWS][p_pad] written by the backward chain ----
constexpr int GRD_ROW_Y0 = 0;                       // dY of trunk layers 0..7 (pre-activation grads), 256 rows each
constexpr int GRD_ROW_SIG = 8 * 256;                // 2048: d sigma_pre (1 row used of 32)   (d bottleneck is never formed: eonerf_pack.h)
constexpr int GRD_ROW_A2 = GRD_ROW_SIG + 32;        // 2080: d albedo_pre (3 rows used of 32)
constexpr int GRD_ROW_A1 = GRD_ROW_A2 + 32;         // 2112: dY albedo hidden (128) ...
constexpr int GRD_ROW_T1 = GRD_ROW_A1 + 128;        // 2240: ... and dY T1 (128) share ONE 256-row block (both multiply X8: one weight-gradient
                                                    //       job, the bottleneck factors), then dY T2..T4 (3 x 128)
constexpr int GRD_ROW_T5 = GRD_ROW_T1 + 512;        // 2752: d {ts_pre, tb_pre} (2 rows used of 32)
constexpr int GRD_ROWS_FULL = GRD_ROW_T5 + 32;      // 2784
constexpr int GRD_ROWS_DENSITY = GRD_ROW_SIG + 32;  // 2080

// row blocks of the two slabs (block-major layout, see eonerf_common.h): the operand units of the weight-gradient jobs
struct ActMap {
    __host__ __device__ static constexpr SlabBlk block(int row) {
        if (row < ACT_ROW_X1) return SlabBlk{ACT_ROW_ENC, 64};
        if (row < ACT_ROW_A1) return SlabBlk{ACT_ROW_X1 + (row - ACT_ROW_X1) / 256 * 256, 256};     // X1..X8
        if (row < ACT_ROW_EMB) return SlabBlk{ACT_ROW_A1 + (row - ACT_ROW_A1) / 128 * 128, 128};    // A1, T1..T4
        return SlabBlk{ACT_ROW_EMB, 32};
    }
};
struct GrdMap {
    __host__ __device__ static constexpr SlabBlk block(int row) {
        if (row < GRD_ROW_SIG) return SlabBlk{row / 256 * 256, 256};                                 // dY0..dY7
        if (row < GRD_ROW_A2) return SlabBlk{GRD_ROW_SIG, 32};
        if (row < GRD_ROW_A1) return SlabBlk{GRD_ROW_A2, 32};
        if (row < GRD_ROW_T1 + 128) return SlabBlk{GRD_ROW_A1, 256};                                 // dY A1 | dY T1
        if (row < GRD_ROW_T5) return SlabBlk{GRD_ROW_T1 + (row - GRD_ROW_T1) / 128 * 128, 128};      // dY T2..T4
        return SlabBlk{GRD_ROW_T5, 32};
    }
};

constexpr int MASK_SLOTS_FULL = 13;                 // trunk 0..7, A1, T1..T4
constexpr int MASK_SLOTS_DENSITY = 8;

// LDS slot (one packed weight chunk; grouping: eonerf_common.h)
template <class P> struct FwdSlot { static constexpr int BYTES = chunk_target<P>() * P::UNIT_B + 1024; };

struct MlpFwdArgs {
    const float *px, *py, *pz;     // [p_pad] compact sample positions (SoA)
    const int* simg;               // [p_pad] image index of each sample (transient embedding row)
    const float* emb;              // [n_img][4] transient_encoder.weight (fp32, read from the flat parameter buffer)
    const int* n_pts;              // device scalar: number of live samples
    int p_pad;                     // leading dimension of every per-sample array (multiple of 256)
    const uint8_t* stream;         // packed weight stream of this kernel variant
    const ChunkDesc* chunks;
    int n_chunks;
    float *sigma, *albedo, *ts, *tb;   // outputs: sigma[p_pad], albedo[3][p_pad], ts[p_pad], tb[p_pad]
    void* act;                     // TRAIN: [ACT_ROWS][p_pad] of P::act_t
    uint32_t* masks;               // TRAIN: [MASK_SLOTS][p_pad][2][4] ReLU masks
    int mask_from;                 // TRAIN: first mask slot the backward will read (the pipelined trunk backward derives ReLU' from the saved
                                   // activations themselves: slots below are not written)
    int* range_flag;               // fp16 x 3 policy: set to 1 when an activation or a position leaves fp16's range (eonerf_range_status)
    int stagger;                   // wave stagger of the 8-wave chains (run_layer): 0 off, 1 waves 4..7 late, 2 odd waves late (A/B control)
};

struct MlpBwdArgs {
    const int* n_pts;
    int p_pad;
    const uint8_t* stream;
    const ChunkDesc* chunks;
    int n_chunks;
    const float *sigma, *albedo, *ts, *tb;      // forward outputs (activation derivatives)
    const float *g_sigma, *g_albedo, *g_ts, *g_tb;   // upstream grads per sample (same SoA shapes)
    const uint32_t* masks;
    void* grd;                     // [GRD_ROWS][p_pad] of P::act_t (A operand of the weight-gradient GEMM)
    float* g_emb;                  // FULL: [p_pad][4] grad wrt the per-sample transient embedding
    const float *px, *py, *pz;     // INPUT_GRAD: positions (encoder derivative)
    float* g_pos;                  // INPUT_GRAD: [3][p_pad]
    uint8_t* dy7_units;            // PIPE 1: dY_7 in B-operand unit order [step of 32 samples][16 KiB]; the rest of the dX chain is left to
                                   // eonerf_bwd_pipe.hip
    int stagger;                   // as MlpFwdArgs::stagger
};

// One weight-gradient GEMM job:  dW[m][col_map[n]] += sum_p  dY^T[m][p] * X^T[n][p]
struct WgradJob {
    const void* a;        // dY^T: first row of the job inside sample tile 0 of the gradient slab
    const void* b;        // X^T : first row of the job inside sample tile 0 of the activation slab
    uint32_t a_stride, b_stride;   // bytes between consecutive sample tiles (= slab rows x 64)
    float* dw;            // destination inside the flat gradient buffer
    float* db;            // bias gradient or nullptr
    float* dw2;           // rows >= split go to a second destination (row - split): two layers that share BOTH operand blocks
    float* db2;           //   (dY A1 | dY T1 against the bottleneck output; d ts_pre | d tb_pre against T4) are one job
    int split, dw2_ld;    // split == m_rows: none
    const int* col_map;   // nullptr = identity; -1 entries are dropped
    const int* n_pts;     // device scalar: live samples of the pass these slabs belong to
    int m_rows, n_rows;   // valid rows of a / b
    int dw_ld;            // row stride of dw
    int gm, gn, wm, wn;   // wave grid and tiles per wave: (gm*wm*32) x (gn*wn*32) >= m_rows x n_rows
    int item0, slices;    // work items [item0, item0 + slices) = equal slices of this job's K range
    int a_units;          // bf16: a's tiles are in B-operand unit order (written by eonerf_bwd_pipe.hip), not feature-major rows
};

// ---- layer-pipelined backward (eonerf_bwd_pipe.hip): 7 stages = trunk layers 7..1 (input dY_7, X images X_7..X_1), floor(CUs / 7) pipelines
constexpr int PIPE_STAGES = 7;        // trunk layers 7..1, one workgroup each
constexpr int PIPE_MAX_STAGES = 7;
constexpr int PIPE_TS = 32;           // samples per step = one sample tile of the bf16 slabs
constexpr int PIPE_RING = 16;         // slots (steps) of an inter-stage ring
constexpr int PIPE_UNIT_B = 16 * 1024;   // one step of a 256-feature tensor in B-operand unit order [k-group 16][lane 64][16 B]
struct BwdPipeArgs {
    const int* n_pts; int p_pad;
    int n_pipes;
    int n_stages;             // PIPE_STAGES
    const uint8_t* wt;        // stage-stationary W^T: [stage][m-tile 8][k-group 16][lane 64][16 B] bf16 (eonerf_pack.cpp)
    const uint8_t* dy_in;     // input of stage 0 of every step in unit order [step][16 KiB] (written by an earlier launch)
    const void* act;          // activation slab (block-major feature-major tiles, eonerf_common.h)
    void* grd;                // gradient slab (trunk launch): the stages of layers 6 and 1 save dY_5 / dY_0 for the remaining GEMM jobs
    uint8_t* rings;           // [pipeline][edge n_stages - 1][PIPE_RING][16 KiB]
    uint32_t* flags;          // [pipeline][edge][64]: head counter at [0], tail counter at [32] (own 128-B lines); zeroed per launch
    uint32_t* scratch_word;   // [workgroup][32]: sink / source of the end stages' fixed-count dummy flag traffic (a line per workgroup)
    int* role_counter;        // zeroed per launch; [0] arrivals, [1..8] arrivals per XCD (xcd_local)
    int xcd_local;            // 1: pipelines are formed INSIDE an XCD wherever 7 workgroups of one XCD exist (roles by HW_REG_XCC_ID behind a
                              // rendezvous of the whole grid): their hand-offs are stored with the default policy -- the tile stays in the XCD's
                              // L2, where the consumer's L1-bypassing loads find it -- instead of write-through to the fabric; what is left
                              // over forms cross-XCD pipelines with the write-through protocol.  Speed only: every edge is correct either way
    int* error;               // the context's STICKY status word: watchdog bits are OR-ed in, never cleared by a launch
    float* d_flat;
    // weight / bias gradient destinations per stage
    size_t dw_off[PIPE_MAX_STAGES], db_off[PIPE_MAX_STAGES];
    int dw_ld[PIPE_MAX_STAGES];
    unsigned long long* stamps;   // diagnostics (EONERF_PIPE_STAMPS): [workgroup role][2 waves][8] cycle sums, or nullptr
    float* partials;          // deterministic mode: [pipeline][stage][256 x 256 dW | 256 db] instead of the atomic flush (reduced in pipeline order)
    int fault_stage;          // test hook (EONERF_PIPE_FAULT): this stage never publishes its tiles -> every watchdog downstream must fire; -1 = off
    // Workgroups beyond the 7 x n_pipes stage roles (256 CUs: 4 of them idle for the whole launch otherwise) run the per-ray ambient-head
    // backward of the step (eonerf_ambient_dev.h), amb_blocks of them; 0: none.  Independent of the stages: nothing waits for them.
    int amb_blocks;
    AmbientBwdArgs amb;
};
hipError_t eo_launch_bwd_pipe(const BwdPipeArgs& a, hipStream_t st);
// input-gradient tail of a pipelined density pass (eonerf_ig_tail.hip)
struct IgTailArgs {
    const int* n_pts; int p_pad;
    const void* grd;          // gradient slab holding the saved dY_0 / dY_5 rows
    const uint8_t* wt;        // [source 2: W_0^T, W_5 skip^T][m-tile 2][k-group 16][lane 64][16 B] bf16 (eonerf_pack.cpp)
    const float *px, *py, *pz;
    float* g_pos;             // [3][p_pad]
};
hipError_t eo_launch_ig_tail(const IgTailArgs& a, int n_wg, hipStream_t st);
// the same tail AND the shadow pass' two weight-gradient products against the encoding in one pass over dY_0 / dY_5 (eonerf_enc_pair.hip)
struct EncPairArgs {
    const int* n_pts; int p_pad;
    const void* grd;          // gradient slab: the dY_0 / dY_5 tiles in unit order (written by the pipelined launch)
    const void* act;          // activation slab: the 64 encoding rows of the pass
    const uint8_t* wt;        // as IgTailArgs::wt
    const float *px, *py, *pz;
    float* g_pos;             // [3][p_pad]
    // The weight-gradient accumulators leave the launch as PARTIALS, one per workgroup (plain 128-B-segment stores): 256 workgroups adding
    // 2 x 256 x 64 floats each into the same 64 + 80 KB of the gradient buffer with atomics took 0.116 ms of a 0.21-ms launch (the memory-side
    // atomic units serialise on so small a footprint, scripts/enc_pair_ablate.sh); k_step_tail sums them at the end of the backward.
    float* part;              // [workgroup][ENC_PART_F]: [source 2][row 256][slot 64] dW | [256] db_0
};
hipError_t eo_launch_enc_pair(const EncPairArgs& a, int n_wg, hipStream_t st);
size_t eo_bwd_pipe_lds_bytes();
bool eo_bwd_pipe_fits_a_cu();

hipError_t eo_launch_mlp_fwd(const MlpFwdArgs& a, int prec, bool full, int mode, int grid, hipStream_t st);      // prec: 0 fp32, 1 bf16, 2 fp16 x 3 (inference)
// pipe: 0 = the whole dX chain; 1 = stop at dY_7 (the trunk is pipelined)
hipError_t eo_launch_mlp_bwd(const MlpBwdArgs& a, bool bf16, bool full, bool input_grad, bool transient, int grid, hipStream_t st, int pipe = 0);
constexpr int WGRAD_MAX_JOBS = 32;     // <= 31 used (fp32 chain + GEMM path, full model); the table must fit the 4-KiB kernel-argument segment
static_assert(sizeof(WgradJob) * WGRAD_MAX_JOBS + 8 + 64 <= 4096, "job table exceeds the kernel-argument segment");
// Riders of ONE job (the bottleneck-factor job [dY_A1; dY_T1] x X_8 of the camera pass): two tiny products whose big operand that job
// streams anyway, so that nothing reads it a second time --
//   sigma row : dW_sigma = sum_p d sigma_pre[p] X_8[:, p]: a ONE-row A operand (a2: row 0 of the d sigma_pre block) against the job's own
//               B fragments, one more MFMA per B fragment on the waves of the first row group (formerly a job of its own: X_8 re-read);
//   embedding : dW_T1[:, 256:260] = sum_p dY_T1[:, p] emb[:, p]^T: the 4 embedding rows (b2) sit beside the ones column of the bias
//               MFMA's B fragment -- columns 1..4 of a product that is computed anyway (formerly a job of its own: dY_T1 re-read).
struct WgradAux {
    int job;              // index in the table, -1: none
    int emb_row0, emb_ld; // rows >= emb_row0 of the job are dY_T1; row stride of dw_emb
    uint32_t a2_stride, b2_stride;
    const void* a2;       // d sigma_pre block, sample tile 0 (nullptr: no sigma rider)
    const void* b2;       // embedding block, sample tile 0 (nullptr: no embedding rider)
    float *dw_sig, *db_sig, *dw_emb;
};
struct WgradJobTable { WgradJob j[WGRAD_MAX_JOBS]; WgradAux aux; int n; int items; };   // by value in the kernel-argument segment; jobs sorted heaviest first
// partials != nullptr (deterministic mode): every work item stores its tile to partials[item] ([256][256] dW | [256] db) instead of
// adding it atomically, and a second kernel sums the items of a job in slice order
// The streaming roles of the pipelined camera launch (eonerf_bwd_pipe.hip) get the job table of the GEMM that follows as a second
// kernel argument: the pipelined path leaves at most 11 jobs (camera: bottleneck factors, albedo output, 3 transient layers, transient
// outputs, layer 0 and skip columns against the encoding; shadow pass: layer 0, skip columns, sigma row), so a short table fits the
// 4-KiB kernel-argument segment beside BwdPipeArgs.
constexpr int WGRAD_STREAM_JOBS = 16;
struct WgradJobTableS { WgradJob j[WGRAD_STREAM_JOBS]; WgradAux aux; int n; int items; };
struct PipeStreamArgs {
    int blocks;               // workgroups beyond the stage (and ambient) roles that run GEMM items; 0: none (the table is not read)
    int ready_items;          // items [0, ready_items) read operands that are final before this launch starts
    int* queue;               // the GEMM's work queue (zeroed with the sync block of the backward's first pipelined launch)
    int* stop;                // [2] the streaming roles' clock: steps run / to run by the first pipeline's first stage (zeroed with the sync block)
    WgradJobTableS tab;
};
static_assert(sizeof(BwdPipeArgs) + sizeof(PipeStreamArgs) <= 4096, "pipelined launch: arguments exceed the kernel-argument segment");
hipError_t eo_launch_bwd_pipe_stream(const BwdPipeArgs& a, const PipeStreamArgs& s, hipStream_t st);
constexpr int WGRAD_PART_F = 256 * 256 + 256;
hipError_t eo_launch_wgrad(const WgradJobTable& jobs, int n_wg, int p_pad, int* queue, bool bf16, hipStream_t st, float* partials = nullptr,
                           bool zero_queue = true);
hipError_t eo_launch_pipe_reduce(const BwdPipeArgs& a, hipStream_t st);
