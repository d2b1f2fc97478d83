// Device side of the weight-gradient GEMM (eonerf_wgrad.hip): the work loop as a device function, so that the k_wgrad launch AND the
// streaming roles of the pipelined trunk backward (eonerf_bwd_pipe.hip: workgroups beyond the stage roles of the camera launch run ready
// GEMM items under the MFMA-bound stages, which leave half of the HBM bandwidth unused) share one body.
#pragma once
#include "eonerf_common.h"
#include "eonerf_kernels.h"
#include <type_traits>

// Diagnostic builds only (scripts/wgrad_ablate.sh): EO_WG_ABL bit 0 drops the atomic flush of an item's tile, bit 1 the MFMAs, bit 2 the LDS-DMA
// of the operands (the counted waits then wait for nothing), bit 3 the per-step barrier.  Results are WRONG with any bit set.
#ifndef EO_WG_ABL
#define EO_WG_ABL 0
#endif
#ifndef EO_WG_DESC
#define EO_WG_DESC 1
#endif

#define EO_WG_MMA(a, b, c) ((EO_WG_ABL & 2) ? (c) : P::mma(a, b, c))

namespace eo_wgrad {

constexpr int WG_NT = 512;
constexpr int MAX_ROWS = 256;
constexpr int ROW_B = SEG_B;                     // bytes per LDS row per K step = one slab segment (bf16: 32 samples, fp32: 16)
constexpr int OPND_B = MAX_ROWS * ROW_B;         // 16 KiB per operand per ring slot
constexpr int SLOT_B = 2 * OPND_B;               // A + B
constexpr int NS = 4;                            // ring slots (128 KiB)
constexpr int DEPTH = NS - 1;                    // K steps in flight ahead of the one being multiplied
constexpr int WMAX = 4, NMAX = 2;
constexpr int AUX_B = 2048;                      // per ring slot: 16 rows x 64 B of the sigma rider's block | of the embedding rider's (WgradAux)
constexpr int AUX_SIG_WAVE = 7, AUX_EMB_WAVE = 6;      // who copies them (the waves with the fewest pieces when the job has < 256 rows)

// 16-B chunk c of row `row` lives at chunk position c ^ ((row >> 2) & 3): a ds_read_b128 lane group (16 rows, one
// chunk) then covers all 64 banks exactly once
EO_DEV int wg_swz(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 16; }

constexpr int WG_SMEM = NS * SLOT_B + NS * AUX_B + 16;

// The work loop of the weight-gradient GEMM: this workgroup (WG_NT threads, WG_SMEM bytes of LDS at `smem`) pulls (job, K slice) items from
// `queue` until none is left.  Tab: WgradJobTable (k_wgrad) or WgradJobTableS (the streaming roles of the pipelined launch).
// BOUNDED (streaming roles of eonerf_bwd_pipe.hip): only items below `ready_items` are claimed (compare-and-swap: an item is never taken
// and dropped) -- the jobs behind them read operands the surrounding launch is still writing -- and an item is only claimed while it
// will be done before the launch's stages are: progress[0] = steps the first pipeline's first stage has run (published every 16 steps),
// progress[1] = the steps it has to run; the role measures its own items (s_memrealtime) and stops claiming when the last one took
// longer than the stages have left at their rate so far.  What is left goes to the k_wgrad launch that follows, on all CUs.
template <class P, class Tab, bool BOUNDED>
EO_DEV void wgrad_work(const Tab& tab, int* queue, float* partials, uint8_t* smem, int ready_items, const int* progress) {
    const unsigned long long t_role = BOUNDED ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long t_claim = t_role, item_ticks = 0;
    typedef typename P::U U;
    constexpr int BK = ROW_B / P::ACT_BYTES;             // samples per K step
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, r = lane & 31;
    uint8_t* const lds_aux = smem + NS * SLOT_B;
    int* const lds_item = reinterpret_cast<int*>(smem + NS * SLOT_B + NS * AUX_B);

  // Work items = (job, slice of the job's K steps), heaviest jobs first; persistent workgroups pull them from one
  // global counter, so slow (latency-bound, few-row) jobs and fast ones balance without any host-side cost model.
  for (;;) {
    __syncthreads();                                    // previous item fully consumed (LDS ring + lds_item)
    if (tid == 0) {
        if constexpr (BOUNDED) {
            int got = -1;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_claim != t_role) item_ticks = now - t_claim;      // the item this role has just finished
            const long long done = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long total = __hip_atomic_load(progress + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool go = true;
            if (total > 0) {      // (not yet published: the launch has only just started)
                // stages' time left at their rate so far: (total - done) x elapsed / done; the first item's duration is not known yet:
                // assume a sixth of the launch
                const long long elapsed = (long long)(now - t_role);
                const long long need = item_ticks ? (long long)item_ticks + (long long)(item_ticks >> 3) : 0;
                if (done >= total) go = false;
                else if (done > 0) go = (total - done) * elapsed > (need ? need : elapsed * total / (6 * done)) * done;
            }
            t_claim = now;
            if (go) {
                int cur = __hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (cur < ready_items) {
                    const int seen = atomicCAS(queue, cur, cur + 1);
                    if (seen == cur) { got = cur; break; }
                    cur = seen;
                }
            }
            *lds_item = got;
        } else {
            *lds_item = atomicAdd(queue, 1);
        }
    }
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(*lds_item);
    if (BOUNDED ? item < 0 : item >= tab.items) return;
    int ji = 0;
    while (ji + 1 < tab.n && item >= tab.j[ji + 1].item0) ++ji;      // <= 40 jobs, uniform scalar scan
    const WgradJob job = tab.j[ji];
    const int w = item - job.item0, n_slices = job.slices;
    const int n_pts = *job.n_pts;
    const int n_pad = (n_pts + P::TILE - 1) / P::TILE * P::TILE;
    const int steps = n_pad / BK;
    const int s0 = (int)((long long)w * steps / n_slices), s1 = (int)((long long)(w + 1) * steps / n_slices);
    if (s0 >= s1) continue;

    const int wm = job.wm, wn = job.wn, gn = job.gn;
    const bool active = wid < job.gm * gn;
    const int wm_idx = wid / gn, wn_idx = wid % gn;

    // ---- LDS-DMA staging: wave `wid` owns rows [32 wid, 32 wid + 32) of both operand tiles; one
    //      buffer_load_dwordx4 ... lds moves 16 rows x 64 B.  Rows past the valid count re-read the last valid row:
    //      they only feed output rows/columns that are never flushed, and every wave issues the same number of loads
    //      per step, which is what the counted vmcnt below relies on. ----
    int voff_a[2], voff_b[2];
    bool on_a[2], on_b[2];          // wave-uniform: this wave's 16-row half-blocks that hold valid rows (the others are never fetched:
    int n_dma = 0;                  // they only feed output rows / columns that are never flushed)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 32 * wid + 16 * j + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const int ra = row < job.m_rows ? row : job.m_rows - 1, rb = row < job.n_rows ? row : job.n_rows - 1;
        // a_units: the A operand of this job is a 16-KiB tile in B-operand unit order (the layer-pipelined backward wrote dY_5 / dY_0
        // that way, eonerf_bwd_pipe.hip): the tile is copied as it is, the fragments come out of it through transposed reads
        voff_a[j] = job.a_units ? (32 * wid + 16 * j) * SEG_B + lane * 16 : ra * SEG_B + chunk * 16;
        voff_b[j] = rb * SEG_B + chunk * 16;
        on_a[j] = 32 * wid + 16 * j < job.m_rows;
        on_b[j] = 32 * wid + 16 * j < job.n_rows;
        n_dma += (on_a[j] ? 1 : 0) + (on_b[j] ? 1 : 0);
    }
    const bool has_aux = tab.aux.job == ji, has_sig = has_aux && tab.aux.a2 != nullptr, has_emb = has_aux && tab.aux.b2 != nullptr;
    const bool dma_sig = has_sig && wid == AUX_SIG_WAVE, dma_emb = has_emb && wid == AUX_EMB_WAVE;
    n_dma += (dma_sig ? 1 : 0) + (dma_emb ? 1 : 0);
    // K step s = sample tile s of the slabs: one contiguous rows x 64 B region per operand.
    // ONE descriptor per operand for the whole item (EO_WG_DESC=1): the step enters through the scalar offset of the load -- a descriptor
    // per step is a 64-bit add and four scalar moves per operand and step (the GEMM spends 11 % of its wave cycles issuing scalar
    // instructions, profiles/r05_b_pmc_sq.csv).  Rows are clamped by the per-lane offsets, so the descriptors need no bounds; step x stride
    // stays below 2^32 inside the size guard of the training entry points (slab_blocks_addressable).
#if EO_WG_DESC
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(job.a)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(job.b)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(has_sig ? tab.aux.a2 : job.a)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(has_emb ? tab.aux.b2 : job.b)), 0, -1, 0x00020000);
#endif
    auto issue = [&](int step, int slot, auto aux_c) {
        if constexpr (decltype(aux_c)::value) {      // (compiled into the Riders loops only: the other loops keep their instruction count)
#if EO_WG_DESC
        if (dma_sig) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)(lds_aux + slot * AUX_B), 16, lane * 16, (uint32_t)step * tab.aux.a2_stride, 0, 2);
        if (dma_emb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b2, (__attribute__((address_space(3))) void*)(lds_aux + slot * AUX_B + 1024), 16, lane * 16, (uint32_t)step * tab.aux.b2_stride, 0, 2);
#else
        if (dma_sig) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(tab.aux.a2)) + (size_t)step * tab.aux.a2_stride, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_aux + slot * AUX_B), 16, lane * 16, 0, 0, 2);
        }
        if (dma_emb) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(tab.aux.b2)) + (size_t)step * tab.aux.b2_stride, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_aux + slot * AUX_B + 1024), 16, lane * 16, 0, 0, 2);
        }
#endif
        }
        if (EO_WG_ABL & 4) return;
        uint8_t* base = smem + slot * SLOT_B + (32 * wid) * ROW_B;
#if EO_WG_DESC
        const uint32_t so_a = (uint32_t)step * job.a_stride, so_b = (uint32_t)step * job.b_stride;
#else
        const uint32_t so_a = 0, so_b = 0;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(job.a)) + (size_t)step * job.a_stride, 0, job.m_rows * SEG_B, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(job.b)) + (size_t)step * job.b_stride, 0, job.n_rows * SEG_B, 0x00020000);
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (on_a[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(base + 16 * j * ROW_B), 16, voff_a[j], so_a, 0, 2);
            if (on_b[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(base + OPND_B + 16 * j * ROW_B), 16, voff_b[j], so_b, 0, 2);
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

    // per-lane LDS offsets of the operand fragments (row, k-group) -> swizzled chunk
    int off_a[WMAX][2], off_b[NMAX][2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
        for (int i = 0; i < WMAX; ++i) {
            const int row = (wm_idx * wm + i) * 32 + r;
            off_a[i][kg] = row * ROW_B + wg_swz(row, 2 * kg + h);
            if (job.a_units) {      // units 2 mt, 2 mt + 1 hold features 32 mt ..; the 16-lane group / lane split of ds_read_b64_tr_b16 (see eonerf_bwd_pipe.hip)
                const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
                off_a[i][kg] = (2 * (wm_idx * wm + i) + (g4 & 1)) * 1024 + ((pp & 1) * 32 + 8 * (g4 >> 1) + qq) * 16 + (pp >> 1) * 8 + 256 * kg;
            }
        }
#pragma unroll
        for (int j = 0; j < NMAX; ++j) { const int row = (wn_idx * wn + j) * 32 + r; off_b[j][kg] = OPND_B + row * ROW_B + wg_swz(row, 2 * kg + h); }
    }

    // the loop is instantiated per number of LDS-DMA pieces this wave issues per step (the counted vmcnt needs an immediate);
    // waves of one workgroup may run different instantiations, they all meet at the same s_barrier once per step
    // sigma rider: every wave takes ONE of its column tiles (tile wm_idx of its wn: the wave grid of the factor job is 2 x 4 with wn = 2,
    // so the two row groups share the columns between them) -- row 0 of `acc` = dW_sigma of those 32 columns; live only in the Riders loop
    struct SigRider { f32x16 acc; float b; };
    const bool sig_here = has_sig && wm_idx < wn;
    // The loop body is instantiated per TILE SHAPE of a wave as well (SHAPE = 16 wm + wn, 0 = run-time shape): with run-time bounds every
    // MFMA and fragment read of a step sits behind a scalar branch (53 per step), and the step is instruction-issue bound, not HBM-bound
    // (a K step costs about the same whatever the job moves; 40 more instructions in it cost 6 % of the launch)
    auto k_loop = [&](auto n_dma_c, auto units_c, auto aux_c, auto shape_c, SigRider* sr) {
        constexpr int N_DMA = decltype(n_dma_c)::value;
        constexpr bool UNITS = decltype(units_c)::value;
        constexpr bool AUX = decltype(aux_c)::value;
        constexpr int SHAPE = decltype(shape_c)::value;
        const int wm = SHAPE ? (SHAPE >> 4) : job.wm, wn = SHAPE ? (SHAPE & 15) : job.wn;      // (shadow the run-time shape)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(s0 + d < s1 ? s0 + d : s1 - 1, d, aux_c);
        for (int s = s0; s < s1; ++s) {
            const int slot = (s - s0) & (NS - 1);
            // this wave's share of step s has landed once at most (DEPTH-1) younger steps are outstanding; the barrier then
            // (a) publishes every wave's share and (b) retires all reads of the slot refilled next
            if (EO_WG_ABL & 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA * (DEPTH - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N_DMA * (DEPTH - 1)) : "memory");
            issue(s + DEPTH < s1 ? s + DEPTH : s1 - 1, (slot + DEPTH) & (NS - 1), aux_c);
            if (active) {
                const uint8_t* T = smem + slot * SLOT_B;
                if constexpr (UNITS) {
                    // A tile in unit order (always the 256 x 64 shape, two m-tiles per wave): the four fragments of the step come out of
                    // eight transposed reads issued back to back behind ONE wait (inline asm: for the intrinsic the wait-count pass
                    // assumes aliasing with the LDS-DMA in flight and drains it)
                    u32x2 t[8];
                    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                                 "ds_read_b64_tr_b16 %1, %8 offset:64\n\t"
                                 "ds_read_b64_tr_b16 %2, %9\n\t"
                                 "ds_read_b64_tr_b16 %3, %9 offset:64\n\t"
                                 "ds_read_b64_tr_b16 %4, %8 offset:256\n\t"
                                 "ds_read_b64_tr_b16 %5, %8 offset:320\n\t"
                                 "ds_read_b64_tr_b16 %6, %9 offset:256\n\t"
                                 "ds_read_b64_tr_b16 %7, %9 offset:320\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                                 : "v"((uint32_t)(uintptr_t)(T + off_a[0][0])), "v"((uint32_t)(uintptr_t)(T + off_a[1][0])) : "memory");
#pragma unroll
                    for (int kg = 0; kg < 2; ++kg) {
                        const U bf0 = lds_unit<P>(T + off_b[0][kg]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const U af = __builtin_bit_cast(U, u32x4{t[4 * kg + 2 * i][0], t[4 * kg + 2 * i][1], t[4 * kg + 2 * i + 1][0], t[4 * kg + 2 * i + 1][1]});
                            acc[i][0] = EO_WG_MMA(af, bf0, acc[i][0]);
                            if (do_bias && i == wn_idx) accb = EO_WG_MMA(af, ones, accb);
                        }
                    }
                } else {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    U af[WMAX], bf[NMAX];
#pragma unroll
                    for (int i = 0; i < WMAX; ++i) if (i < wm) af[i] = lds_unit<P>(T + off_a[i][kg]);
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) if (j < wn) bf[j] = lds_unit<P>(T + off_b[j][kg]);
                    U baux = ones;
                    if constexpr (AUX) {
                        const uint8_t* X = lds_aux + slot * AUX_B + (2 * kg + h) * 16;
                        if (has_emb && r >= 1 && r <= 4) baux = lds_unit<P>(X + 1024 + (r - 1) * ROW_B);      // columns 1..4 = the embedding rows
                        if (sig_here) {      // wave-uniform
                            U as = P::zero();
                            if (r == 0) as = lds_unit<P>(X);                                                    // row 0 = d sigma_pre
                            sr->acc = EO_WG_MMA(as, wm_idx == 0 ? bf[0] : bf[NMAX - 1], sr->acc);
                            if (wid == 0) {
#pragma unroll
                                for (int e = 0; e < P::NE; ++e) sr->b += (float)as[e];
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < WMAX; ++i)
                        if (i < wm) {
#pragma unroll
                            for (int j = 0; j < NMAX; ++j)
                                if (j < wn) acc[i][j] = EO_WG_MMA(af[i], bf[j], acc[i][j]);
                            if (do_bias && i == wn_idx) accb = EO_WG_MMA(af[i], baux, accb);
                        }
                }
                }
            }
        }
    };
    typedef std::integral_constant<bool, false> Rows;
    typedef std::integral_constant<bool, P::IS_BF16> Units;
    typedef std::integral_constant<bool, false> Plain;
    typedef std::integral_constant<bool, true> Riders;
    typedef std::integral_constant<int, 0> AnyShape;
#define EO_KL(N, U, A, S, SR) k_loop(std::integral_constant<int, N>(), U(), A(), std::integral_constant<int, S>(), SR)
    const int shape = 16 * wm + wn;
    if (job.a_units) {      // A = a whole 256-row tile (two pieces per wave), B = the 64 encoding rows (two pieces on waves 0 and 1) or the
                            // 4 embedding rows (one piece on wave 0)
        if (n_dma == 4) EO_KL(4, Units, Plain, 0, nullptr);
        else if (n_dma == 3) EO_KL(3, Units, Plain, 0, nullptr);
        else EO_KL(2, Units, Plain, 0, nullptr);
    } else if (has_aux) {      // the bottleneck-factor job with its riders: 256 x 256 (4 x 2 tiles per wave) or 128 x 256 (2 x 2)
        SigRider sr{zero_acc(), 0.f};
        if (shape == 0x42 && n_dma == 4) EO_KL(4, Rows, Riders, 0x42, &sr);
        else if (shape == 0x42 && n_dma == 5) EO_KL(5, Rows, Riders, 0x42, &sr);
        else if (shape == 0x22 && n_dma == 4) EO_KL(4, Rows, Riders, 0x22, &sr);
        else if (shape == 0x22 && n_dma == 3) EO_KL(3, Rows, Riders, 0x22, &sr);
        else if (shape == 0x22 && n_dma == 2) EO_KL(2, Rows, Riders, 0x22, &sr);
        else switch (n_dma) {
            case 2: EO_KL(2, Rows, Riders, 0, &sr); break;
            case 3: EO_KL(3, Rows, Riders, 0, &sr); break;
            case 4: EO_KL(4, Rows, Riders, 0, &sr); break;
            default: EO_KL(5, Rows, Riders, 0, &sr); break;
        }
        if (sig_here && active) {
            if (h == 0) {      // accumulator element 0 of the lower lane half = output row 0
                const int col = (wn_idx * wn + (wm_idx == 0 ? 0 : NMAX - 1)) * 32 + r;
                if (col < job.n_rows) atomicAdd(tab.aux.dw_sig + col, sr.acc[0]);
            }
            if (wid == 0 && r == 0) atomicAdd(tab.aux.db_sig, sr.b);      // lanes 0 and 32: the two halves of every K group
        }
    }
    // the shapes and piece counts the job tables of eonerf_api.hip produce ...
    else if (shape == 0x42 && n_dma == 4) EO_KL(4, Rows, Plain, 0x42, nullptr);       // 256 x 256
    else if (shape == 0x22 && n_dma == 4) EO_KL(4, Rows, Plain, 0x22, nullptr);       // 128 x 256, waves 0-3
    else if (shape == 0x22 && n_dma == 2) EO_KL(2, Rows, Plain, 0x22, nullptr);       //            waves 4-7
    else if (shape == 0x21 && n_dma == 4) EO_KL(4, Rows, Plain, 0x21, nullptr);       // 128 x 128, waves 0-3
    else if (shape == 0x21 && n_dma == 0) EO_KL(0, Rows, Plain, 0x21, nullptr);       //            waves 4-7
    else if (shape == 0x11 && n_dma == 3) EO_KL(3, Rows, Plain, 0x11, nullptr);       // few-row jobs (sigma row, output layers, embedding columns)
    else if (shape == 0x11 && n_dma == 2) EO_KL(2, Rows, Plain, 0x11, nullptr);
    else if (shape == 0x11 && n_dma == 0) EO_KL(0, Rows, Plain, 0x11, nullptr);
    else switch (n_dma) {                                                              // ... and anything else
        case 0: EO_KL(0, Rows, Plain, 0, nullptr); break;
        case 1: EO_KL(1, Rows, Plain, 0, nullptr); break;
        case 2: EO_KL(2, Rows, Plain, 0, nullptr); break;
        case 3: EO_KL(3, Rows, Plain, 0, nullptr); break;
        default: EO_KL(4, Rows, Plain, 0, nullptr); break;
    }
#undef EO_KL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the tail prefetches before the LDS is released

    // ---- flush: fp32 atomics, 32 consecutive columns per half-wave instruction ----
    if (active && !(EO_WG_ABL & 1)) {
#pragma unroll
    for (int i = 0; i < WMAX; ++i)
        if (i < wm) {
#pragma unroll
            for (int j = 0; j < NMAX; ++j)
                if (j < wn) {
                    const int col = (wn_idx * wn + j) * 32 + r;
                    int cm = col;
                    if (col < job.n_rows && job.col_map) cm = job.col_map[col];
                    if (partials) {      // deterministic mode: the tile as it stands ([row][256 columns]), reduced by k_wgrad_reduce
                        if (col < job.n_rows) {
                            float* pt = partials + (size_t)item * WGRAD_PART_F;
#pragma unroll
                            for (int g = 0; g < 16; ++g) {
                                const int row = (wm_idx * wm + i) * 32 + acc_row(g, h);
                                if (row < job.m_rows) pt[row * 256 + col] = acc[i][j][g];
                            }
                        }
                    } else if (col < job.n_rows && cm >= 0) {
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const int row = (wm_idx * wm + i) * 32 + acc_row(g, h);
                            if (row < job.split) { if (job.dw) atomicAdd(job.dw + (size_t)row * job.dw_ld + cm, acc[i][j][g]); }      // (dw == nullptr: these rows have no destination)
                            else if (row < job.m_rows) atomicAdd(job.dw2 + (size_t)(row - job.split) * job.dw2_ld + cm, acc[i][j][g]);
                        }
                    }
                }
        }
    if (do_bias && r == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = (wm_idx * wm + wn_idx) * 32 + acc_row(g, h);
            if (row < job.m_rows) {
                if (partials) partials[(size_t)item * WGRAD_PART_F + 256 * 256 + row] = accb[g];
                else atomicAdd(row < job.split ? job.db + row : job.db2 + (row - job.split), accb[g]);
            }
        }
    }
    // riders (never with `partials`: the deterministic mode keeps them as jobs of their own)
    if (do_bias && has_emb && r >= 1 && r <= 4) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = (wm_idx * wm + wn_idx) * 32 + acc_row(g, h);
            if (row >= tab.aux.emb_row0 && row < job.m_rows) atomicAdd(tab.aux.dw_emb + (size_t)(row - tab.aux.emb_row0) * tab.aux.emb_ld + (r - 1), accb[g]);
        }
    }
    }
  }
}


}  // namespace eo_wgrad
