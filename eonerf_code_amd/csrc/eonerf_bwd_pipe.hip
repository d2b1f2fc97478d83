// Layer-pipelined trunk backward (bf16): the dX chain AND the weight-gradient GEMM of trunk layers 7..1 in ONE launch, with the
// inter-layer gradients dY_l kept on chip (H10 of SURVEY.md 8a; the layers are radiance_fields/mlp.py:87-101).
//
// The chain kernel (eonerf_mlp_bwd.hip) walks ALL layers for its 256 samples, so it re-streams every weight per tile and has to
// park every dY_l in HBM for a separate GEMM (eonerf_wgrad.hip): 2.3 GB written + 5.4 GB read per 4096-ray step.  Here the
// roles are turned around: one workgroup per LAYER ("stage"), 7 stages = one pipeline, floor(CUs / 7) pipelines.
//   * W_l^T (128 KB bf16) is STATIONARY in the registers of the stage (wave w owns output m-tile w: 16 A units = 64 VGPRs);
//   * dW_l (256 x 256 fp32) is STATIONARY in registers for the whole launch (wave w owns rows 32w..32w+31: 128 VGPRs);
//   * per step of 32 samples the stage takes dY_l from its input ring (B-operand "unit" order [k-group 16][lane 64][16 B],
//     exactly what the producer's accumulators pack to) and the layer input X_l from the activation slab (feature-major rows ARE the
//     B fragments of the dW product, and X_l > 0 is the ReLU' predicate), both by LDS-DMA into a 4-slot LDS ring, 3 steps in flight;
//       dX = W_l^T dY_l                      16 MFMAs per wave  (A registers, B ds_read_b128 of the dY image)
//       dY_{l-1} = dX .* (X_l > 0)           packed to two 1-KiB units, stored write-through (sc1) into the output ring
//       dW_l += dY_l X_l^T                   16 MFMAs per wave  (A = ds_read_b64_tr_b16 of the SAME dY image, B = X image)
//       db_l += row sums of dY_l             VALU on the A fragments
//   * dY_6 .. dY_1 never touch HBM: the rings (16 slots x 16 KiB per edge) live in the Infinity Cache and are overwritten in place.
//     The two gradients the remaining GEMM jobs need against the encoding (dY_5: skip columns of layer 5, dY_0: layer 0) are written
//     ONCE, in the same unit order, into their tiles of the gradient slab: the layer-6 stage's "ring" towards layer 5 IS that linear
//     buffer (never overwritten, so no back-pressure on this edge), the layer-1 stage just stores there; eonerf_wgrad.hip reads these
//     two operands through transposed LDS reads (WgradJob::a_units), eonerf_ig_tail.hip takes them as B units as they are.
// Hand-off protocol (placement independent, cdna guide G16 / R1): payload stores sc1, every storing wave's stores are known
// complete through the counted vmcnt of a LATER step's barrier, then ONE lane stores the edge's `head` counter (agent scope);
// the consumer's control wave polls `head` (sc1 loads, one step ahead of use) and loads the payload with sc1 LDS-DMA; it
// returns ring slots through the edge's `tail` counter the same way.  Every spin is bounded by a wall-clock watchdog
// (s_memrealtime): on expiry the stage raises *error, tells its workgroup through LDS and leaves -- the launch always drains.
// Residency: grid = 7 x pipelines <= CU count, one workgroup per CU (LDS), roles taken from an arrival counter, so every pipeline
// that processes samples is complete as soon as its last workgroup is scheduled.
#include "eonerf_common.h"
#include "eonerf_kernels.h"
#include "eonerf_ambient_dev.h"
#include "eonerf_wgrad_dev.h"

// Diagnostic builds only (scripts/pipe_ablate.sh): EO_PABL bit 0 drops the dW MFMAs, bit 1 the dX MFMAs, bit 2 the B-fragment LDS reads of
// both products, bit 3 the LDS-DMA refill, bit 4 the ring / slab stores, bit 5 the epilogue's VALU work and the bias sums, bit 6 the final
// flush of the stationary gradients; bit 7 (128) drops the WHOLE weight-gradient phase (MFMAs, fragment reads, bias sums: what a dX-only
// half-stage would run), bit 8 (256) the whole dX phase (MFMAs, fragment reads, ReLU', packing, output stores; the flags still flow: a
// dW-only half-stage).  Results are WRONG with any bit set; the shipped library is built with EO_PABL == 0.
// EO_COR = 1 (round 6, scripts/coresidency.sh): the co-residency falsifier -- the stage kernel is capped at 128 registers
// (__launch_bounds__(512, 4): two such workgroups' worth of waves per CU) and runs with three LDS slots (96 KB, prefetch distance 2), so
// that a second workgroup (k_cor_partner: a pure LDS-DMA stream, 4 waves, 48 KB of LDS) fits beside it on the same CU.
#ifndef EO_PABL
#define EO_PABL 0
#endif
#ifndef EO_COR
#define EO_COR 0
#endif

namespace {

constexpr int NT = 512;
constexpr int TS = PIPE_TS;                       // samples per step
constexpr int IMG_B = 16 * 1024;                  // one step of a 256-feature tensor (bf16)
constexpr int SLOT_B = 2 * IMG_B;                 // dY image | X image
// EO_PIPE_DW16 = 1: the weight-gradient product runs on v_mfma_f32_16x16x32_bf16 (one K step = the 32 samples of a pipeline step; the
// wave's 32 x 256 block = 2 x 16 tiles of 16 x 16), 0 (default): on v_mfma_f32_32x32x16_bf16 like the dX product.  Same MFMA cycles and
// LDS bytes; measured on the same box (parity green): 1.4 % fewer cycles per step, 1 % MORE wall time for the kernel (0.848 vs 0.839 ms)
// -- the higher sustained clock the guide reports for the 16x16x32 shape in MFMA-bound loops does not show in this 50 %-busy loop.
#ifndef EO_PIPE_DW16
#define EO_PIPE_DW16 0
#endif
#ifndef EO_PIPE_DEPTH
#define EO_PIPE_DEPTH (EO_COR ? 2 : 3)
#endif
#ifndef EO_PIPE_ORDB
#define EO_PIPE_ORDB 0
#endif
#ifndef EO_PIPE_SPREAD
#define EO_PIPE_SPREAD 0
#endif
// Experiment builds of the XCD-local pipelines (scripts/xcd_ab.sh): hand-off stores of an intra-XCD edge with the default policy (1) or
// write-through like a cross-XCD edge (0); its loads sc1 (0) or streaming (1); ring slots in use (a power of two <= PIPE_RING)
#ifndef EO_XCD_PLAIN
#define EO_XCD_PLAIN 1
#endif
#ifndef EO_XCD_NT
#define EO_XCD_NT 0
#endif
#ifndef EO_RING_USE
#define EO_RING_USE PIPE_RING
#endif
// EO_PIPE_EARLY_TR (round 6): bit 0 -- the transposed A-fragment reads of the dW phase are issued in FRONT of the step's DMA issue block
// (their LDS latency runs under the block instead of in front of the first dW MFMA); bit 1 -- the ReLU' reads of the dX phase are issued
// three MFMAs before the end of its chain instead of behind it.
#ifndef EO_PIPE_EARLY_TR
#define EO_PIPE_EARLY_TR 1      // bit 0 on since round 6: -1.9 % on the camera launch, -0.55 % on the step (profiles/r06_early_tr_reads.txt); bit 1: neutral
#endif
#ifndef EO_PIPE_XM_AT      // (bit 1: behind which MFMA of the dX chain the ReLU' reads are issued)
#define EO_PIPE_XM_AT 14
#endif
// EO_PIPE_DMA03 = 1 (round 6 experiment): the LDS-DMA pieces of a step are all issued by waves 0-3 (eight each: their own and their SIMD
// partner's), waves 4-7 -- the critical path of a step (profiles/r06_pipe_stamps_ablation.txt) -- issue none.
#ifndef EO_PIPE_DMA03
#define EO_PIPE_DMA03 0
#endif
// EO_PIPE_DDEPTH (round 6): prefetch distance of the dY image alone (the X image keeps EO_PIPE_DEPTH).  The dY tile of a step comes from the
// producer stage through the Infinity Cache, the X image from HBM; what a stage fetches DDEPTH steps ahead its producer must have published
// DDEPTH steps ahead, so every step of dY distance is a step of lag on each of the six edges of a pipeline (fill + drain: 9 % of a launch).
// 2 since round 6: -1.4 % / -1.0 % on the camera / shadow-pass launch against 3 (profiles/r06_dy_prefetch_distance.txt).  A distance of 1
// (the DMA block issued first thing in a step, one step to land; profiles/r06_dy_distance_1_experiment.patch) is 5 % SLOWER: the load does not make it.
#ifndef EO_PIPE_DDEPTH
#define EO_PIPE_DDEPTH (EO_PIPE_DMA03 || EO_PIPE_SPREAD || EO_COR ? EO_PIPE_DEPTH : 2)
#endif
#ifndef EO_PIPE_STAMPS      // 1: the per-phase cycle stamps of scripts/pipe_stamps.py are compiled in (scripts/stamp.sh builds that library)
#define EO_PIPE_STAMPS 0
#endif
constexpr int NSLOT = EO_COR ? 3 : 4, DEPTH = EO_PIPE_DEPTH;      // LDS ring slots; steps of DMA in flight ahead of the one being multiplied (<= NSLOT - 1)
EO_DEV int slot_of(int k) { return (NSLOT & (NSLOT - 1)) == 0 ? (k & (NSLOT - 1)) : k % NSLOT; }
static_assert(DEPTH >= 2 && DEPTH <= NSLOT - 1, "prefetch distance");
constexpr int DDEPTH = EO_PIPE_DDEPTH;
static_assert(DDEPTH >= 2 && DDEPTH <= DEPTH && (DDEPTH == DEPTH || (!EO_PIPE_DMA03 && !EO_PIPE_SPREAD && !EO_COR)), "dY prefetch distance");
constexpr uint32_t RING_USE = EO_RING_USE;
static_assert(RING_USE <= PIPE_RING && (RING_USE & (RING_USE - 1)) == 0, "ring slots in use");
constexpr int N_DMA = 4;                          // LDS-DMA pieces per wave per step: 2 dY + 2 X
constexpr int CTRL_B = 64;
constexpr int SMEM_B = NSLOT * SLOT_B + CTRL_B;
constexpr int AUX_SC1 = 16, AUX_NT = 2;
constexpr unsigned long long WATCHDOG_TICKS = 30000000ull;     // 0.3 s of the 100 MHz s_memrealtime clock

typedef __attribute__((address_space(1))) unsigned int gu32;

EO_DEV int wg_swz16(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 16; }      // eonerf_wgrad.hip's ring swizzle

// ops a wave issues per step, in program order: [CTRL: 2 flag stores, 2 flag polls] .. 2 payload stores .. N_DMA pieces
// second order (waves 4..7): N_DMA pieces .. 2 payload stores
constexpr int NST = 2;
template <bool CTRL, bool ORDB> struct Cnt {
    static constexpr int ND = EO_PIPE_DMA03 ? (ORDB ? 0 : 2 * N_DMA) : N_DMA;      // pieces THIS wave issues per step
    static constexpr int C = (CTRL ? 4 : 0) + NST + ND;
    // top of step s: the stores of step s-2 are complete (=> publishable), hence also the polls of step s-2 and the DMA of step s.
    // First order: the pieces of step s-2 and all of step s-1 are younger than those stores; second order: only step s-1
    // (first order, DMA side: the pieces of step s were issued DEPTH steps ago behind that step's stores, so DEPTH - 1 whole steps are
    //  younger; the stricter of the two conditions counts)
    // (DDEPTH < DEPTH, first order: the dY pieces of step s were issued DDEPTH steps ago behind that step's stores, in front of its X pieces:
    //  with DDEPTH = 2 the X pieces of step s-2 and all of step s-1 are younger -- ND / 2 + C; the stores of step s-2 are older.  Second order
    //  and control wave: unchanged, the dY pieces of step s are older than what their conditions already wait for)
    static constexpr int TOP = ORDB ? C : (DDEPTH < DEPTH ? ND / 2 + (DDEPTH - 1) * C : ((DEPTH - 1) * C < ND + C ? (DEPTH - 1) * C : ND + C));
    // control wave (first order): tighter -- only the payload stores and the pieces of step s-1 stay outstanding, so the flag polls
    // of step s-1 are in (one step of latency instead of two: every stage then runs one step closer behind its producer)
    static constexpr int TOP_CTRL = NST + ND;
};

struct Stage {
    int pipe, st, n_k;
    bool has_in;               // false: stage 0, its input comes from an earlier launch (a.dy_in)
    int x_row;                 // first row of the 256-row block of the activation slab this stage reads as its X image
    const uint8_t* in_lin;     // != nullptr: the input tiles lie in a linear buffer [global step][16 KiB] written inside this launch (layer 5: dY_5 in the gradient slab)
    uint8_t* out_lin;          // MODE 1 / 2: the linear output buffer [global step][16 KiB]
    bool local;                // every stage of this pipeline runs on ONE XCD (BwdPipeArgs::xcd_local)
    int* done;                 // != nullptr (first stage of the first pipeline of a launch with streaming roles): [0] = steps run so far (every
                               // 16th step and at the end), [1] = steps to run -- the streaming roles' clock (eonerf_wgrad_dev.h)
};

// MODE 0: the output goes to the next stage's ring; 1 (layer 6): the output goes, write-through like a ring slot, to its tile of the
// dY_5 block of the gradient slab, which the layer-5 stage reads as its "ring" and the skip-column GEMM job reads later; 2 (layer 1): the
// output dY_0 goes to its tile of the dY_0 block (streaming stores, nobody in this launch reads it).
template <bool CTRL, int MODE, bool ORDB = false>
EO_DEV void run_stage(const BwdPipeArgs& a, const Stage& S, uint8_t* smem, int tid) {
    constexpr bool HAS_OUT = MODE == 0 || MODE == 1;   // a consumer inside this launch (flags)
    constexpr bool RING_OUT = MODE == 0;       // ... whose ring slots come back through the tail counter
    typedef PBf16 P;
    typedef P::U U;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c = lane & 31;
    int* const ctl = reinterpret_cast<int*>(smem + NSLOT * SLOT_B);      // [0] abort, [4..7] polled head, [8..11] polled tail
    const int n_k = S.n_k;

    // ---- stationary operands ----
    U wt[16];
    {
        const uint8_t* src = a.wt + ((size_t)(S.st * 8 + wave) * 16) * 1024 + lane * 16;
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) wt[kg] = *reinterpret_cast<const U*>(src + kg * 1024);
        // an explicit wait the compiler's wait-count pass SEES: without it the pass treats these loads as possibly outstanding
        // inside the loop and puts its own (stricter) vmcnt in front of the first MFMAs of every step
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    }
#if EO_PIPE_DW16
    f32x4 dw[2][16];       // [m-subtile of 16 rows][n-subtile of 16 columns]: row 4 (lane >> 4) + reg, column lane & 15
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < 16; ++ns) dw[ms][ns] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db[2] = {0.f, 0.f};
#else
    f32x16 dw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dw[j] = zero_acc();
    float db = 0.f;
#endif

    // ---- sources ----
    const size_t n_tiles = (size_t)a.p_pad / TS;                       // sample tiles of the slabs
    const int n_edges = a.n_stages - 1;
    const uint8_t* x_base = reinterpret_cast<const uint8_t*>(a.act) + (size_t)S.x_row * n_tiles * SEG_B;     // block start (rows x_row.., tile 0)
    uint8_t* const ring_in = a.rings + (size_t)(S.pipe * n_edges + (S.st - 1)) * PIPE_RING * IMG_B;
    uint8_t* const ring_out = a.rings + (size_t)(S.pipe * n_edges + S.st) * PIPE_RING * IMG_B;
    gu32* const f_in = (gu32*)(a.flags + (size_t)(S.pipe * n_edges + (S.st - 1)) * 64);          // [0] head, [32] tail
    gu32* const f_out = (gu32*)(a.flags + (size_t)(S.pipe * n_edges + S.st) * 64);
    // dummy flag traffic of the end stages (fixed op counts) goes to a line of the workgroup's own: 36 control waves storing to and
    // polling ONE shared line every step serialise at the memory side and stall the whole CU's vector-memory issue
    gu32* const my_scratch = (gu32*)(a.scratch_word + (size_t)(S.pipe * a.n_stages + S.st) * 32);

    // linear unit-order buffers (one 16-KiB tile per 32 samples): this stage's output (MODE 1, 2: its tile of a 256-row block of the
    // gradient slab, same footprint as the feature-major tile the chain + GEMM path keeps there), its input (layer 5)
    uint8_t* const grd_blk = S.out_lin;
    const uint8_t* const in_blk = S.in_lin;

    // per-lane DMA source offsets of the X image: wave w stages rows 32w..32w+31, 16 rows per piece, chunks XOR-swizzled
    constexpr int NDW = Cnt<CTRL, ORDB>::ND;      // (EO_PIPE_DMA03: waves 0-3 also stage the rows / units of waves 4-7)
    int x_voff[NDW > 4 ? 4 : 2];
#pragma unroll
    for (int j = 0; j < (NDW > 4 ? 4 : 2); ++j) {
        const int row = 32 * (wave + 4 * (j >> 1)) + 16 * (j & 1) + (lane >> 2);
        x_voff[j] = row * SEG_B + (((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
    // DMA of this pipeline's k-th step (clamped by the caller) into LDS slot k & 3: N_DMA pieces of 1 KiB per wave, issued in a
    // block of their own between the phases.  In-kernel stamps (round 3) put that block at 320 cycles per step on waves 0..3 and 530 on
    // waves 4..7 -- the critical path of a step -- so the pieces were also issued ONE AT A TIME between the MFMAs of the dW phase
    // (build switch EO_PIPE_SPREAD=1, same vmcnt order, parity green): 1 % SLOWER on the same box (3.82 vs 3.79 ms full, 2.104 vs 2.090
    // rgb).  Left off: the block form is what the partner wave's matrix phase overlaps best.
    // ONE descriptor per source for the whole launch; the step enters through the scalar offset of the load (a descriptor per step
    // cost ~40 scalar instructions and a handful of branches in every step of every wave: the stage is instruction-issue bound --
    // 320 instructions per wave and step around 32 MFMAs)
    const uint32_t lin_stride = (uint32_t)a.n_pipes * IMG_B;      // this pipeline's consecutive steps in a linear buffer (sample tile = pipe + k n_pipes)
    const bool in_ring = S.has_in && !in_blk;
    const uint8_t* const d_base = !S.has_in ? a.dy_in + (size_t)S.pipe * IMG_B : (in_blk ? in_blk + (size_t)S.pipe * IMG_B : ring_in);
    const uint32_t d_mask = in_ring ? RING_USE - 1 : 0x7fffffffu, d_mul = in_ring ? IMG_B : lin_stride;
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(d_base), 0, -1, 0x00020000);
    // X image: rows of the activation slab, written by the forward kernel of an earlier launch (streaming: nt)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(x_base) + (size_t)S.pipe * 256 * SEG_B, 0, -1, 0x00020000);
    // output: the next stage's ring (MODE 0) or this pipeline's tiles of a linear buffer
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(MODE == 0 ? ring_out : grd_blk + (size_t)S.pipe * IMG_B, 0, -1, 0x00020000);
    const int v_dy = lane * 16 + (2 * wave) * 1024;               // this wave's two pieces of a 16-KiB unit-order image (loads and stores)
    struct Dma { uint32_t so_d, so_x; uint8_t *slot, *slot_d; bool on; };      // slot: of the X image's step, slot_d: of the dY image's step
    auto dma_prep = [&](int kx, int kd) {
        Dma d;
        d.on = !((EO_PABL & 8) && kx >= DEPTH);
        d.slot = smem + slot_of(kx) * SLOT_B;
        d.slot_d = smem + slot_of(kd) * SLOT_B;
        d.so_d = ((uint32_t)kd & d_mask) * d_mul;
        d.so_x = (uint32_t)kx * lin_stride;
        return d;
    };
    auto dma_piece = [&](const Dma& d, int i4) {      // i4: compile-time constant at every call site; pieces 4..7: the partner wave's share
        if (!d.on) return;
        const int i = i4 & 3, vwave = wave + 4 * (i4 >> 2), v_dy = lane * 16 + (2 * vwave) * 1024, wave = vwave;
        const int* x_voff_ = x_voff + 2 * (i4 >> 2);
        if (i < 2) {
            // handed over inside this launch: sc1; the first stage's input comes from an earlier launch: streaming
            if (EO_XCD_NT && S.has_in && S.local)      // (experiment build: streaming instead of sc1 loads on an intra-XCD edge)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(d.slot_d + (2 * wave + i) * 1024), 16,
                                                         v_dy + i * 1024, d.so_d, 0, AUX_NT);
            else if (S.has_in)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(d.slot_d + (2 * wave + i) * 1024), 16,
                                                         v_dy + i * 1024, d.so_d, 0, AUX_SC1);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (__attribute__((address_space(3))) void*)(d.slot_d + (2 * wave + i) * 1024), 16,
                                                         v_dy + i * 1024, d.so_d, 0, AUX_NT);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(d.slot + IMG_B + (32 * wave + 16 * (i - 2)) * SEG_B), 16,
                                                     x_voff_[i - 2], d.so_x, 0, AUX_NT);
        }
    };
    auto issue = [&](int k, bool with_dy) {      // all pieces in one block (prologue)
        const Dma d = dma_prep(k, k);
#pragma unroll
        for (int i = 0; i < NDW; ++i) if (with_dy || (i & 3) >= 2) dma_piece(d, i);
    };

    // ---- per-lane LDS read offsets ----
    // A fragments of the dW product: transposed reads of the dY image (see the file header): 16-lane group g4, lane i = 4qq + pp
    const int g4 = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int tr_off = (2 * wave + (g4 & 1)) * 1024 + ((pp & 1) * 32 + 8 * (g4 >> 1) + qq) * 16 + (pp >> 1) * 8;     // + 64 t + 256 ks
    // B fragments of the dW product: row 32j + c of the X image, 16-byte chunk 2ks + h
    const int xb_off0 = IMG_B + c * SEG_B + wg_swz16(c, h), xb_off1 = IMG_B + c * SEG_B + wg_swz16(c, 2 + h);        // + 2048 j
#if EO_PIPE_DW16
    // 16x16x32 operands.  A (lane = feature row lane & 15 of the m-subtile, K chunk lane >> 4 = samples 8 g4 .. 8 g4 + 7): two transposed
    // reads (4 samples each) of unit 2 wave + ms; lane 4 qq + pp of the group supplies sample 8 g4 + qq (+ 4), features 4 pp .. 4 pp + 3
    // of the unit = lane half pp & 1, elements 4 (pp >> 1) ..  B (lane = feature row lane & 15 of the n-subtile, same K chunk): one
    // 16-byte chunk g4 of row 16 ns + (lane & 15) of the X image; the swizzle class (row >> 2) & 3 does not depend on ns
    const int tr16_off = (2 * wave) * 1024 + ((pp & 1) * 32 + 8 * g4 + qq) * 16 + (pp >> 1) * 8;           // + 1024 ms, + 64 for samples + 4
    const int xb16_off = IMG_B + i16 * SEG_B + wg_swz16(i16, g4);                                           // + 1024 ns
#endif
    // ReLU' of layer - 1 = (X_layer > 0) on the bf16 values the forward saved (the same predicate its mask bits record): this
    // lane's 16 (feature, sample) pairs of the dX accumulator come out of the X image through 4 transposed reads -- 16-lane group
    // g4 = samples 16 (g4 & 1) .., half h = g4 >> 1; read q covers rows 32 wave + 8q + 4h .. +3 (one swizzle class per read)
    int xm_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wave + 8 * q + 4 * (g4 >> 1);
        xm_off[q] = IMG_B + (f0 + qq) * SEG_B + wg_swz16(f0, 2 * (g4 & 1) + (pp >> 1)) + (pp & 1) * 8;
    }

    // ---- control state (wave 0) ----
    int known_head = S.has_in ? 0 : 0x7fffffff, known_tail = RING_OUT ? 0 : 0x7fffffff;
    unsigned ph = 0, pt = 0;          // poll results in flight (inline asm: the compiler must not see these as loads)
    auto poll_sync = [&](gu32* word) -> int {      // slow path: one synchronous agent-scope read
        return (int)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // bounded spin (watchdog).  Once this workgroup has given up (its own watchdog, or another stage's: the status word is polled in
    // the slow path), every later wait returns at once: the launch drains in a handful of steps instead of one 0.3 s timeout per
    // wait and stage.  (The status word is sticky across launches: until the host has read it, every pipelined launch whose
    // hand-offs ever enter the slow path leaves early too -- its gradients are not used anyway, see k_adam.)
    bool aborted = false;
    auto wait_for = [&](gu32* word, int need, int& known) {
        if (known >= need || aborted) return;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned it = 0;; ++it) {
            known = poll_sync(word);
            if (known >= need) return;
            __builtin_amdgcn_s_sleep(2);
            const bool expired = __builtin_amdgcn_s_memrealtime() - t0 > WATCHDOG_TICKS;
            const bool other = (it & 63) == 63 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (expired || other) {
                if (lane == 0) { if (expired) atomicOr(a.error, 1 << (S.st & 7)); ctl[0] = 1; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                aborted = true;
                return;
            }
        }
    };

    // diagnostics: cycle sums of this wave (total loop, slow-path spins, counted wait, barrier, dX phase, dW phase)
    const bool stamp = EO_PIPE_STAMPS && a.stamps != nullptr;
    auto clk = [&]() -> uint32_t { return (uint32_t)__builtin_amdgcn_s_memtime(); };      // (32-bit sums: 64-bit ones cost the stamp build spilled registers, and a spill's reload drains the DMA queue)      // (build switch: scripts/stamp.sh; the production loop carries no stamp code)
    uint32_t t_slow = 0, t_top = 0, t_bar = 0, t_dx = 0, t_is = 0, t_dw = 0, n_slow = 0;
    const uint32_t t_begin = stamp ? clk() : 0;

    // ---- prologue: first DEPTH steps in flight ----
    if (CTRL && S.has_in) wait_for(f_in, n_k < DDEPTH ? n_k : DDEPTH, known_head);
    if (CTRL) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (ctl[0]) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d < n_k ? d : n_k - 1, d < DDEPTH);      // (X images DEPTH steps ahead, dY images DDEPTH)
    // step 0 only: nothing but the other two prologue steps is younger than its pieces (the loop's counted wait assumes the
    // steady state, where two whole steps of stores and pieces are)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NDW - (DEPTH - DDEPTH) * (NDW / 2)) : "memory");

    if (CTRL && lane == 0 && S.done) __hip_atomic_store(S.done + 1, n_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if EO_PABL      // (diagnostic variants: a shorter body must not be unrolled into a different register budget)
#pragma clang loop unroll(disable)
#endif
    for (int k = 0; k < n_k; ++k) {
        uint8_t* slot = smem + slot_of(k) * SLOT_B;
        if (CTRL && S.done && (k & 15) == 15 && lane == 0) __hip_atomic_store(S.done, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- top of the step ----
        const uint32_t tt0 = stamp ? clk() : 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CTRL ? Cnt<CTRL, ORDB>::TOP_CTRL : Cnt<CTRL, ORDB>::TOP) : "memory");
        const uint32_t tt1 = stamp ? clk() : 0;
        const bool half = (EO_PABL & 512) && (k & 1);      // (diagnostic: barrier and flag work on every second step only -- results are garbage, timing only)
        if (CTRL && !half) {
            // the flag values polled in the previous step have landed behind the counted wait
            asm volatile("" : "+v"(ph), "+v"(pt));
            if (k >= 1) {
                const int vh = __builtin_amdgcn_readfirstlane((int)ph), vt = __builtin_amdgcn_readfirstlane((int)pt);
                if (S.has_in && vh > known_head) known_head = vh;
                if (RING_OUT && vt > known_tail) known_tail = vt;
            }
            // make sure (slow path only when the pipeline is starved or backed up) that what this step needs exists
            const int need_in = (k + DDEPTH + ((EO_PABL & 512) ? 1 : 0) < n_k ? k + DDEPTH + ((EO_PABL & 512) ? 1 : 0) : n_k - 1) + 1;       // tiles that must be published for this step's DMA
            const int need_out = k + 1 - RING_USE;                                   // tiles the consumer must have released
            const uint32_t ts0 = stamp ? clk() : 0;
            const bool slow = (S.has_in && known_head < need_in) || (RING_OUT && known_tail < need_out);
            if (S.has_in && known_head < need_in) wait_for(f_in, need_in, known_head);
            if (RING_OUT && known_tail < need_out) wait_for(f_out + 32, need_out, known_tail);
            if (stamp && slow) { t_slow += clk() - ts0; ++n_slow; }
        }
        if (!half) asm volatile("s_barrier" ::: "memory");
        const uint32_t tt2 = stamp ? clk() : 0;
        if (stamp) { t_top += tt1 - tt0; t_bar += tt2 - tt1; }
        // a watchdog fired in this workgroup: every wave leaves behind the same barrier (looked at every 8th step: the LDS round trip
        // costs every wave ~100 cycles, and a stalled pipeline is in no hurry)
        if ((k & 7) == 0 && ctl[0]) return;
        if (CTRL && !half) {
            // behind the barrier: every wave's share of step k has landed (=> the ring slot of tile k can go back) and every
            // wave's stores of step k-2 are complete (=> tiles 0..k-2 are published)
            if (lane == 0) {
                if (S.has_in) __hip_atomic_store(f_in + 32, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(my_scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (HAS_OUT) __hip_atomic_store(f_out, (unsigned)(k > 0 && S.st != a.fault_stage ? k - 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store((my_scratch + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // polls for step k+2 (always two loads: fixed op count)
            const gu32* a_h = S.has_in ? f_in : my_scratch;
            const gu32* a_t = RING_OUT ? f_out + 32 : my_scratch;
            asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1"
                         : "=&v"(ph), "=&v"(pt) : "v"(a_h), "v"(a_t) : "memory");
        }

        // ---- the three phases of a step; waves 4..7 (ORDB) run them in another order than waves 0..3, so that on every SIMD one
        //      wave's matrix phase meets its partner's LDS / VALU / vector-memory phase (after the barrier both would otherwise
        //      multiply at the same time and then both idle the matrix pipe at the same time) ----
        auto phase_dx = [&]() {        // dX = W^T dY (16 MFMAs, B units of the dY image 3 reads ahead), ReLU', pack, hand on
            f32x16 acc = zero_acc();
            u32x2 xm[4] = {};
            {
                const uint8_t* bp = slot + lane * 16;
                // operand window: WIN reads ahead of the MFMA that consumes them
                constexpr int WIN = 3;
                U fr[WIN];
#pragma unroll
                for (int d = 0; d < WIN; ++d) fr[d] = lds_unit<P>(bp + d * 1024);
#pragma unroll
                for (int kg = 0; kg < 16; ++kg) {
                    if (!(EO_PABL & 2)) acc = P::mma(wt[kg], fr[kg % WIN], acc);
                    if (!(EO_PABL & 4) && kg + WIN < 16) fr[kg % WIN] = lds_unit<P>(bp + (kg + WIN) * 1024);
                    if ((EO_PIPE_EARLY_TR & 2) && kg == EO_PIPE_XM_AT)      // the last B unit has been requested: the window's registers free up from here on
                        asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                                     "ds_read_b64_tr_b16 %1, %5\n\t"
                                     "ds_read_b64_tr_b16 %2, %6\n\t"
                                     "ds_read_b64_tr_b16 %3, %7"
                                     : "=&v"(xm[0]), "=&v"(xm[1]), "=&v"(xm[2]), "=&v"(xm[3])
                                     : "v"((uint32_t)(uintptr_t)(slot + xm_off[0])), "v"((uint32_t)(uintptr_t)(slot + xm_off[1])),
                                       "v"((uint32_t)(uintptr_t)(slot + xm_off[2])), "v"((uint32_t)(uintptr_t)(slot + xm_off[3])) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (EO_PIPE_EARLY_TR & 2)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xm[0]), "+v"(xm[1]), "+v"(xm[2]), "+v"(xm[3]) :: "memory");
            else
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                         "ds_read_b64_tr_b16 %1, %5\n\t"
                         "ds_read_b64_tr_b16 %2, %6\n\t"
                         "ds_read_b64_tr_b16 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(xm[0]), "=&v"(xm[1]), "=&v"(xm[2]), "=&v"(xm[3])
                         : "v"((uint32_t)(uintptr_t)(slot + xm_off[0])), "v"((uint32_t)(uintptr_t)(slot + xm_off[1])),
                           "v"((uint32_t)(uintptr_t)(slot + xm_off[2])), "v"((uint32_t)(uintptr_t)(slot + xm_off[3])) : "memory");
            uint32_t w8[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {      // word s = accumulator registers 2s, 2s+1 <-> activation word s of the transposed reads
                if (EO_PABL & 32) { w8[s] = xm[s >> 1][s & 1]; continue; }
                uint32_t flags, r;
                const uint32_t xw = xm[s >> 1][s & 1];
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(flags) : "v"(xw), "v"(0x00010001u));        // post-ReLU bf16 >= 0: 1 where > 0
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(cvt_pk_bf16(acc[2 * s], acc[2 * s + 1])), "v"(flags));
                w8[s] = r;
            }
            if (!(EO_PABL & 16)) {
                const uint32_t so_o = MODE == 0 ? ((uint32_t)k & (RING_USE - 1)) * IMG_B : (uint32_t)k * lin_stride;
                if (EO_XCD_PLAIN && (MODE == 0 || MODE == 1) && S.local) {      // read inside this launch by a workgroup of the SAME XCD: default policy, the
                    // lines stay in the shared L2 (the consumer's sc1 loads bypass its L1 only); complete -- at the L2 -- behind the same
                    // counted vmcnt as the write-through form
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[0], w8[1], w8[2], w8[3]}, rs_o, v_dy, so_o, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[4], w8[5], w8[6], w8[7]}, rs_o, v_dy + 1024, so_o, 0);
                } else if (MODE == 0 || MODE == 1) {      // read inside this launch, possibly from another XCD: write-through
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[0], w8[1], w8[2], w8[3]}, rs_o, v_dy, so_o, AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[4], w8[5], w8[6], w8[7]}, rs_o, v_dy + 1024, so_o, AUX_SC1);
                } else {      // read by a LATER launch (the GEMM's jobs; the trunk launch): streaming
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[0], w8[1], w8[2], w8[3]}, rs_o, v_dy, so_o, AUX_NT);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w8[4], w8[5], w8[6], w8[7]}, rs_o, v_dy + 1024, so_o, AUX_NT);
                }
            }
        };
#if !EO_PIPE_DW16
        u32x2 ta[2][2] = {};
        auto dw_prefetch = [&]() {
            if (!(EO_PIPE_EARLY_TR & 1) || (EO_PABL & 128)) return;
            const uint32_t ra = (uint32_t)(uintptr_t)(slot + tr_off);
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                         "ds_read_b64_tr_b16 %1, %4 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %4 offset:256\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:320"
                         : "=&v"(ta[0][0]), "=&v"(ta[0][1]), "=&v"(ta[1][0]), "=&v"(ta[1][1]) : "v"(ra) : "memory");
        };
#else
        auto dw_prefetch = [&]() {};
#endif
        auto phase_dw = [&](const Dma& dma) {        // dW += dY X^T over the 32 samples of the step, db += row sums; the next DMA pieces in between
#if EO_PIPE_DW16
            // A fragments = the dY image read TRANSPOSED (inline asm: for the intrinsic the wait-count pass assumes aliasing with the
            // LDS-DMA in flight and drains it)
            u32x2 ta[2][2];
            const uint32_t ra = (uint32_t)(uintptr_t)(slot + tr16_off);
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                         "ds_read_b64_tr_b16 %1, %4 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:1088\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(ta[0][0]), "=&v"(ta[0][1]), "=&v"(ta[1][0]), "=&v"(ta[1][1]) : "v"(ra) : "memory");
            const U af0 = __builtin_bit_cast(U, u32x4{ta[0][0][0], ta[0][0][1], ta[0][1][0], ta[0][1][1]});
            const U af1 = __builtin_bit_cast(U, u32x4{ta[1][0][0], ta[1][0][1], ta[1][1][0], ta[1][1][1]});
            const uint8_t* xb = slot + xb16_off;
            U bf[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) bf[d] = lds_unit<P>(xb + d * 1024);
#pragma unroll
            for (int ns = 0; ns < 16; ++ns) {
                if (!(EO_PABL & 1)) {
                    dw[0][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bf[ns % 3], dw[0][ns], 0, 0, 0);
                    dw[1][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf[ns % 3], dw[1][ns], 0, 0, 0);
                }
                if (!(EO_PABL & 4) && ns + 3 < 16) bf[ns % 3] = lds_unit<P>(xb + (ns + 3) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bias gradient: this lane's 8 samples of features 32 wave + 16 ms + (lane & 15)
            if (!(EO_PABL & 32)) {
                const u32x4 a0 = __builtin_bit_cast(u32x4, af0), a1 = __builtin_bit_cast(u32x4, af1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    db[0] += __uint_as_float(a0[e] << 16) + __uint_as_float(a0[e] & 0xffff0000u);
                    db[1] += __uint_as_float(a1[e] << 16) + __uint_as_float(a1[e] & 0xffff0000u);
                }
            }
#else
            // A fragments = the dY image read TRANSPOSED (inline asm: for the intrinsic the wait-count pass assumes aliasing with the
            // LDS-DMA in flight and drains it)
            const uint32_t ra = (uint32_t)(uintptr_t)(slot + tr_off);
            if (EO_PIPE_EARLY_TR & 1)      // issued by dw_prefetch() in front of the DMA issue block
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0][0]), "+v"(ta[0][1]), "+v"(ta[1][0]), "+v"(ta[1][1]) :: "memory");
            else
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                         "ds_read_b64_tr_b16 %1, %4 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %4 offset:256\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:320\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(ta[0][0]), "=&v"(ta[0][1]), "=&v"(ta[1][0]), "=&v"(ta[1][1]) : "v"(ra) : "memory");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const U af = __builtin_bit_cast(U, u32x4{ta[ks][0][0], ta[ks][0][1], ta[ks][1][0], ta[ks][1][1]});
                const uint8_t* xb = slot + (ks ? xb_off1 : xb_off0);
                U bf[2];
                bf[0] = lds_unit<P>(xb);
                bf[1] = lds_unit<P>(xb + 2048);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (!(EO_PABL & 1)) dw[j] = P::mma(af, bf[j & 1], dw[j]);
                    if (!(EO_PABL & 4) && j + 2 < 8) bf[j & 1] = lds_unit<P>(xb + (j + 2) * 2048);
#if EO_PIPE_SPREAD
                    if ((j & 1) && 4 * ks + (j >> 1) < N_DMA) dma_piece(dma, 4 * ks + (j >> 1));      // pieces 0..3 behind MFMAs 1, 3, 5, 7 of the first K step
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                // bias gradient: this lane's 8 samples of feature 32 wave + (lane & 31)
                const u32x4 av = __builtin_bit_cast(u32x4, af);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(EO_PABL & 32)) db += __uint_as_float(av[e] << 16) + __uint_as_float(av[e] & 0xffff0000u);
            }
#endif
        };
        const int k_next = k + DEPTH < n_k ? k + DEPTH : n_k - 1;       // refill: into the slot step k-1 used (free behind this step's barrier)
        const Dma dma = dma_prep(k_next, k + DDEPTH < n_k ? k + DDEPTH : n_k - 1);      // (dY image: DDEPTH steps ahead, into the dY half of THAT step's slot)
        auto issue_block = [&]() {
#if !EO_PIPE_SPREAD
#pragma unroll
            for (int i = 0; i < NDW; ++i) dma_piece(dma, i);
#endif
        };
        if (!ORDB) {
            if (!(EO_PABL & 256)) phase_dx();
            const uint32_t tt3 = stamp ? clk() : 0;
            dw_prefetch();
            issue_block();
            const uint32_t tt4 = stamp ? clk() : 0;
            if (!(EO_PABL & 128)) phase_dw(dma);
            if (stamp) { t_dx += tt3 - tt2; t_is += tt4 - tt3; t_dw += clk() - tt4; }
        } else {
#if EO_PIPE_ORDB == 1      // DMA issue -> dX -> dW (measured 0.8 % slower, round 3)
            issue_block();
            const uint32_t tt3 = stamp ? clk() : 0;
            phase_dx();
            const uint32_t tt4 = stamp ? clk() : 0;
            phase_dw(dma);
            if (stamp) { t_is += tt3 - tt2; t_dx += tt4 - tt3; t_dw += clk() - tt4; }
#else
            dw_prefetch();
            issue_block();
            const uint32_t tt3 = stamp ? clk() : 0;
            if (!(EO_PABL & 128)) phase_dw(dma);
            const uint32_t tt4 = stamp ? clk() : 0;
            if (!(EO_PABL & 256)) phase_dx();
            if (stamp) { t_is += tt3 - tt2; t_dw += tt4 - tt3; t_dx += clk() - tt4; }
#endif
        }
    }
    if (stamp && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)(S.pipe * a.n_stages + S.st) * 8 + wave) * 16;
        o[0] = clk() - t_begin; o[1] = t_slow; o[2] = t_top; o[3] = t_bar; o[4] = t_dx; o[5] = n_slow; o[6] = (unsigned long long)n_k;
        o[7] = 0; o[8] = t_is; o[9] = 0; o[10] = t_dw;
    }
    if (CTRL && lane == 0 && S.done) __hip_atomic_store(S.done, n_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- drain: the last stores become visible, the last tiles are published ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (CTRL && lane == 0 && HAS_OUT && S.st != a.fault_stage) __hip_atomic_store(f_out, (unsigned)n_k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- flush the stationary gradients: fp32 atomics (one partial per pipeline and element), or, in deterministic mode, plain
    //      stores of this workgroup's partial, summed in pipeline order by k_pipe_reduce ----
#if EO_PIPE_DW16
    // element (row 32 wave + 16 ms + 4 g4 + reg, column 16 ns + i16) = dw[ms][ns][reg]; the bias sums of a feature sit in the four lanes
    // i16, i16 + 16, i16 + 32, i16 + 48 (one K chunk each)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) { db[ms] += __shfl_xor(db[ms], 16, 64); db[ms] += __shfl_xor(db[ms], 32, 64); }
    if (a.partials) {
        float* pt = a.partials + (size_t)(S.pipe * a.n_stages + S.st) * (256 * 256 + 256);
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < 16; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) pt[(32 * wave + 16 * ms + 4 * g4 + g) * 256 + 16 * ns + i16] = dw[ms][ns][g];
        if (g4 == 0) { pt[256 * 256 + 32 * wave + i16] = db[0]; pt[256 * 256 + 32 * wave + 16 + i16] = db[1]; }
        return;
    }
    float* dwp = a.d_flat + a.dw_off[S.st];
    const int ld = a.dw_ld[S.st];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < 16; ++ns)
#pragma unroll
            for (int g = 0; g < 4; ++g) atomicAdd(dwp + (size_t)(32 * wave + 16 * ms + 4 * g4 + g) * ld + 16 * ns + i16, dw[ms][ns][g]);
    if (g4 == 0) {
        atomicAdd(a.d_flat + a.db_off[S.st] + 32 * wave + i16, db[0]);
        atomicAdd(a.d_flat + a.db_off[S.st] + 32 * wave + 16 + i16, db[1]);
    }
#else
    db += __shfl_xor(db, 32, 64);
    if (a.partials) {
        float* pt = a.partials + (size_t)(S.pipe * a.n_stages + S.st) * (256 * 256 + 256);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) pt[(32 * wave + acc_row(g, h)) * 256 + 32 * j + c] = dw[j][g];
        if (h == 0) pt[256 * 256 + 32 * wave + c] = db;
        return;
    }
    if (EO_PABL & 64) return;      // diagnostic: no flush at all (what the 64 K atomics per workgroup cost)
    float* dwp = a.d_flat + a.dw_off[S.st];
    float* dbp = a.d_flat + a.db_off[S.st];
    const int ld = a.dw_ld[S.st];
    const int row0 = 32 * wave;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) atomicAdd(dwp + (size_t)(row0 + acc_row(g, h)) * ld + 32 * j + c, dw[j][g]);
    if (h == 0) atomicAdd(dbp + row0 + c, db);
#endif
}

// role (pipeline, stage) of this workgroup from the arrival counter, step count of its pipeline; false: no stage work -- `extra` >= 0 then
// numbers the workgroups beyond the 7 x n_pipes stage roles (-1: a stage role of a pipeline without samples)
EO_DEV bool take_role(const BwdPipeArgs& a, uint8_t* smem, int tid, Stage& S, int& extra) {
    int* const ctl = reinterpret_cast<int*>(smem + NSLOT * SLOT_B);
    if (tid == 0) {
        ctl[0] = 0; ctl[2] = 0;
        if (!a.xcd_local) ctl[1] = atomicAdd(a.role_counter, 1);
        else {
            // XCD-aware roles.  Every workgroup registers with its XCD's counter, then with the grid's; once the whole grid has (all of
            // it is resident: one workgroup per CU) the eight totals are final and every workgroup derives the SAME assignment from them:
            // XCD x forms floor(c_x / 7) pipelines of its own (arrival rank r -> pipeline r / 7, stage r % 7), the c_x % 7 workgroups
            // left over join the cross-XCD pool, numbered XCD by XCD.  Nothing is assumed about how workgroups are placed; the wait is
            // bounded like every other (watchdog -> status word, the workgroup leaves)
            const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);      // HW_REG_XCC_ID, bits 3:0
            const int rx = atomicAdd(a.role_counter + 1 + xcc, 1);
            __threadfence();
            atomicAdd(a.role_counter, 1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool ok = true;
            while (__hip_atomic_load(a.role_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 > WATCHDOG_TICKS) { ok = false; break; }
            }
            __threadfence();
            if (!ok) { atomicOr(a.error, 1 << 7); ctl[0] = 1; ctl[1] = 0x3fffffff; }
            else {
                int before_local = 0, before_left = 0, n_local = 0, mine = 0;
                for (int x = 0; x < 8; ++x) {
                    const int c = __hip_atomic_load(a.role_counter + 1 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pl = c / a.n_stages;
                    if (x < xcc) { before_local += pl; before_left += c - pl * a.n_stages; }
                    if (x == xcc) mine = pl;
                    n_local += pl;
                }
                if (rx < mine * a.n_stages) { ctl[1] = (before_local + rx / a.n_stages) * a.n_stages + rx % a.n_stages; ctl[2] = 1; }
                else ctl[1] = n_local * a.n_stages + before_left + (rx - mine * a.n_stages);
            }
        }
    }
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(ctl[1]);
    S.pipe = role / a.n_stages; S.st = role % a.n_stages;
    S.local = __builtin_amdgcn_readfirstlane(ctl[2]) != 0;
    extra = -1;
    if (ctl[0]) return false;      // (the rendezvous timed out)
    if (S.pipe >= a.n_pipes) { extra = role - a.n_pipes * a.n_stages; return false; }
    const int n_pts = *a.n_pts;
    // whole 256-sample tiles, as the chain kernels process them (dead samples carry zero gradients): the GEMM jobs that follow read
    // the saved dY_0 / dY_5 rows of every sample tile up to that bound
    const int n_steps = (n_pts + 255) / 256 * (256 / TS);
    S.n_k = S.pipe < n_steps ? (n_steps - S.pipe + a.n_pipes - 1) / a.n_pipes : 0;
    S.has_in = S.st > 0;
    S.in_lin = nullptr; S.out_lin = nullptr; S.done = nullptr;
    return S.n_k > 0;
}

EO_DEV void run_role(const BwdPipeArgs& a, Stage& S, uint8_t* smem, int tid) {
    const int layer = 7 - S.st;
    const size_t n_tiles = (size_t)a.p_pad / TS;
    S.x_row = ACT_ROW_X1 + 256 * (layer - 1);                          // input of layer `layer` = output of layer - 1
    uint8_t* const grd = reinterpret_cast<uint8_t*>(a.grd);
    if (layer == 5) S.in_lin = grd + (size_t)(GRD_ROW_Y0 + 5 * 256) * n_tiles * SEG_B;                         // dY_5 tiles, written by the layer-6 stage
    if (layer == 6 || layer == 1) S.out_lin = grd + (size_t)(GRD_ROW_Y0 + (layer - 1) * 256) * n_tiles * SEG_B;    // dY_5 / dY_0 tiles
    const int wv = tid >> 6;
    // wave 0: control wave (first order); waves 1..3: first order; waves 4..7 (the second wave of every SIMD): second order
    if (layer == 6) {
        if (wv == 0) run_stage<true, 1>(a, S, smem, tid); else if (wv < 4) run_stage<false, 1>(a, S, smem, tid); else run_stage<false, 1, true>(a, S, smem, tid);
    } else if (layer == 1) {
        if (wv == 0) run_stage<true, 2>(a, S, smem, tid); else if (wv < 4) run_stage<false, 2>(a, S, smem, tid); else run_stage<false, 2, true>(a, S, smem, tid);
    } else {
        if (wv == 0) run_stage<true, 0>(a, S, smem, tid); else if (wv < 4) run_stage<false, 0>(a, S, smem, tid); else run_stage<false, 0, true>(a, S, smem, tid);
    }
}

#if EO_COR
__global__ __launch_bounds__(NT, 4) void k_bwd_pipe(BwdPipeArgs a) {
#else
__global__ __launch_bounds__(NT) void k_bwd_pipe(BwdPipeArgs a) {
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    Stage S;
    int extra;
    if (!take_role(a, smem, tid, S, extra)) {
        // no stage left for this workgroup: the launch's spare CUs take the ambient-head backward of the step
        static_assert(NT == 128 * AMB_STREAMS && AMB_LDS_F * 4 <= NSLOT * SLOT_B, "the ambient body runs with a stage workgroup's shape and LDS");
        if (extra >= 0 && extra < a.amb_blocks) ambient_bwd_body(a.amb, extra, a.amb_blocks, reinterpret_cast<float*>(smem));
        return;
    }
    run_role(a, S, smem, tid);
}

// The same launch with STREAMING roles (the camera pass' launch): the stages are MFMA-bound and use about half of the HBM bandwidth, the
// weight-gradient GEMM that follows is HBM-bound -- and most of its jobs read operands that are final before this launch starts (the
// heads' gradients of this pass, everything of the shadow pass).  s.blocks workgroups beyond the stage roles therefore run the GEMM's
// work loop on those items (eonerf_wgrad_dev.h: the same queue, bounded claims) until the first stage-0 workgroup reports its last
// step; the k_wgrad launch that follows takes what is left, including the jobs that read the dY_0 / dY_5 tiles written here.
__global__ __launch_bounds__(NT) void k_bwd_pipe_stream(BwdPipeArgs a, PipeStreamArgs s) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    static_assert(NT == eo_wgrad::WG_NT, "the GEMM's work loop runs with a stage workgroup's shape");
    const int tid = threadIdx.x;
    Stage S;
    int extra;
    if (!take_role(a, smem, tid, S, extra)) {
        if (extra >= 0 && extra < a.amb_blocks) ambient_bwd_body(a.amb, extra, a.amb_blocks, reinterpret_cast<float*>(smem));
        else if (extra >= a.amb_blocks && extra < a.amb_blocks + s.blocks) {
            __syncthreads();      // (take_role's control words share the LDS with the GEMM ring)
            eo_wgrad::wgrad_work<PBf16, WgradJobTableS, true>(s.tab, s.queue, nullptr, smem, s.ready_items, s.stop);
        }
        return;
    }
    if (S.st == 0 && S.pipe == 0) S.done = s.stop;
    run_role(a, S, smem, tid);
}

// deterministic mode: block = (stage, row), thread = column; pipelines that had no step wrote nothing and are skipped
__global__ __launch_bounds__(256) void k_pipe_reduce(BwdPipeArgs a) {
    const int st = blockIdx.y, row = blockIdx.x, col = threadIdx.x;
    const int n_steps = (*a.n_pts + 255) / 256 * (256 / TS);
    const int live = n_steps < a.n_pipes ? n_steps : a.n_pipes;
    float acc = 0.f, accb = 0.f;
    for (int p = 0; p < live; ++p) {
        const float* pt = a.partials + (size_t)(p * a.n_stages + st) * (256 * 256 + 256);
        acc += pt[row * 256 + col];
        if (col == 0) accb += pt[256 * 256 + row];
    }
    a.d_flat[a.dw_off[st] + (size_t)row * a.dw_ld[st] + col] += acc;
    if (col == 0) a.d_flat[a.db_off[st] + row] += accb;
}

#if EO_COR
// The dummy streaming partner of the co-residency falsifier: 4 waves per workgroup, each wave streams its share of `src` through a
// private 12-KB LDS region (3 slots x 4 pieces of 1 KiB) by LDS-DMA with two iterations (8 KB per wave, 32 KB per workgroup) in flight
// and reads nothing back: the transfer rate of a stream that shares its CU with a stage.
constexpr int COR_NT = 256, COR_LDS = 4 * 3 * 4096;
__global__ __launch_bounds__(COR_NT) void k_cor_partner(const uint8_t* src, unsigned long long bytes_per_wg, unsigned long long total) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long base = (unsigned long long)blockIdx.x * bytes_per_wg + (unsigned long long)wave * (bytes_per_wg / 4);
    const int iters = (int)(bytes_per_wg / 4 / 4096);
    uint8_t* mine = smem + wave * 3 * 4096;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long off = (base + (unsigned long long)it * 4096) % (total - 8192);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) + (off & ~15ull), 0, 4096, 0x00020000);
        uint8_t* dst = mine + (it % 3) * 4096;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, lane * 16, p * 1024, 0, AUX_NT);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

}  // namespace

#if EO_COR
hipError_t eo_launch_cor_partner(const void* src, size_t total_bytes, size_t stream_bytes, int n_wg, hipStream_t st) {
    hipLaunchKernelGGL(k_cor_partner, dim3(n_wg), dim3(COR_NT), COR_LDS, st, reinterpret_cast<const uint8_t*>(src),
                       (unsigned long long)(stream_bytes / n_wg / 16384 * 16384), (unsigned long long)total_bytes);
    return hipGetLastError();
}
#endif

hipError_t eo_launch_pipe_reduce(const BwdPipeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_pipe_reduce, dim3(256, a.n_stages), dim3(256), 0, st, a);
    return hipGetLastError();
}

size_t eo_bwd_pipe_lds_bytes() { return SMEM_B; }

// can ONE workgroup of the pipelined kernel be resident on a CU of the current device at all (LDS, registers)?
bool eo_bwd_pipe_fits_a_cu() {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B) != hipSuccess) return false;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bwd_pipe, NT, SMEM_B) != hipSuccess) return false;
    return nb >= 1;
}

hipError_t eo_launch_bwd_pipe_stream(const BwdPipeArgs& a, const PipeStreamArgs& s, hipStream_t st) {
    if (a.n_stages != PIPE_STAGES || a.partials) return hipErrorInvalidValue;
    constexpr int LDS = SMEM_B > eo_wgrad::WG_SMEM ? SMEM_B : eo_wgrad::WG_SMEM;
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_pipe_stream), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bwd_pipe_stream, dim3(a.n_pipes * PIPE_STAGES + a.amb_blocks + s.blocks), dim3(NT), LDS, st, a, s);
    return hipGetLastError();
}

hipError_t eo_launch_bwd_pipe(const BwdPipeArgs& a, hipStream_t st) {
    if (a.n_stages != PIPE_STAGES) return hipErrorInvalidValue;
    static EoAttrOnce attr;
    {
        const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bwd_pipe, dim3(a.n_pipes * PIPE_STAGES + a.amb_blocks), dim3(NT), SMEM_B, st, a);
    return hipGetLastError();
}
