// Common device-side definitions for the EO-NeRF HIP path (gfx950 / CDNA4 only).
//
// Layout vocabulary used by every kernel in this directory
//   sample  : one (ray, interval) pair that survived the cube filter (sat_rendering.py:79-82); samples of a
//             ray are contiguous, rays are in batch order -> "compact" index p in [0, n_pts).
//   tile    : P::TILE consecutive samples handled by one workgroup; one wave owns 32 of them, sample = lane&31.
//   H^T     : activations are kept TRANSPOSED in registers: MFMA rows = features, MFMA columns (lanes) = samples,
//             so layer l+1 = W_l (A operand, streamed through LDS) x H_l^T (B operand, straight from the
//             previous accumulators: no LDS round trip, no lane movement).
//   k-group : the features covered by one 1-KiB A-operand read (ds_read_b128 per lane): 16 features for bf16
//             (one v_mfma_f32_32x32x16_bf16), 8 features for fp32 (four v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define EO_DEV __device__ __forceinline__

// Diagnostic builds only (scripts/ablate.sh): EO_ABL bit 0 drops the per-chunk s_barrier, bit 1 the A-operand LDS re-reads,
// bit 2 the ReLU/mask work of the epilogues, bit 3 the weight prefetch, bit 4 the HBM traffic of the slab stores.  Results are WRONG with any bit set; the shipped
// library is built with EO_ABL == 0.
#ifndef EO_ABL
#define EO_ABL 0
#endif
// EO_STAMP (diagnostic builds only, scripts/stamp.sh): s_memtime stamps around the waits of the chain kernels, summed per wave
// into eo_stamps[] (total, copy wait, barrier wait, flush wait, waves) and read back by eonerf_debug_read().
#ifdef EO_STAMP
#define EO_T() __builtin_amdgcn_s_memtime()
#endif

// row of the 32x32 accumulator tile held in register r by a lane of half h (cdna guide, C/D map)
EO_DEV constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// Precision policies
// ------------------------------------------------------------------------------------------------
struct PBf16 {
    static constexpr bool IS_BF16 = true, IS_SPLIT = false;
    static constexpr int UNIT_B = 1024;      // one A-operand unit (a k-group of one m-tile) in the packed streams and in LDS
    static constexpr int WAVES = 8, NT = 512, TILE = 256;
    static constexpr int KF = 16;            // features per k-group
    static constexpr int KG32 = 2;           // k-groups per 32 features
    static constexpr int NE = 8;             // elements per lane per k-group
    typedef bf16x8 U;                        // per-lane operand unit (A or B) of one k-group
    typedef __bf16 act_t;                    // storage type of saved activations / gradients
    static constexpr int ACT_BYTES = 2;
    __host__ __device__ static constexpr int feat(int kg, int h, int e) { return 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3); }
    EO_DEV static f32x16 mma(const U& a, const U& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    EO_DEV static U zero() { U u; for (int i = 0; i < 8; ++i) u[i] = (__bf16)0.0f; return u; }
};

struct PF32 {
    static constexpr bool IS_BF16 = false, IS_SPLIT = false;
    static constexpr int UNIT_B = 1024;
    static constexpr int WAVES = 4, NT = 256, TILE = 128;
    static constexpr int KF = 8;
    static constexpr int KG32 = 4;
    static constexpr int NE = 4;
    typedef f32x4 U;
    typedef float act_t;
    static constexpr int ACT_BYTES = 4;
    __host__ __device__ static constexpr int feat(int kg, int h, int e) { return 8 * kg + 4 * h + e; }
    EO_DEV static f32x16 mma(const U& a, const U& b, f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    EO_DEV static U zero() { U u = {0.f, 0.f, 0.f, 0.f}; return u; }
};

// Split precision for EXPORT renders (round 4): every operand is carried as hi + lo, two fp16 numbers (11 + 11 significant bits, lo in
// fp16's subnormal range where it has to be: ~2^-21 relative, against 2^-24 for fp32 and 2^-17 for a bf16 pair), and a product is
// three v_mfma_f32_32x32x16_f16 -- hi x hi + hi x lo + lo x hi, fp32 accumulate; lo x lo (2^-22 of the product) is dropped.  3/16 of
// the matrix time of the fp32 path (v_mfma_f32_32x32x2_f32) at fp32-level accuracy on this network (measured against the fp32 oracle:
// tests/test_bf16_fullsize.py).  Inference only: 4 waves x 512 registers like the fp32 policy (an activation costs 32 bits either way).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct F16Pair { f16x8 hi, lo; };
struct PH3 {
    static constexpr bool IS_BF16 = false, IS_SPLIT = true;
    static constexpr int UNIT_B = 2048;      // [hi 1 KiB][lo 1 KiB]
    static constexpr int WAVES = 4, NT = 256, TILE = 128;
    static constexpr int KF = 16;
    static constexpr int KG32 = 2;
    static constexpr int NE = 8;
    typedef F16Pair U;
    typedef float act_t;                     // (never saved: no training in this precision)
    static constexpr int ACT_BYTES = 4;
    __host__ __device__ static constexpr int feat(int kg, int h, int e) { return 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3); }      // as PBf16
    EO_DEV static f32x16 mma(const U& a, const U& b, f32x16 c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);      // the small terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    EO_DEV static U zero() { U u; for (int i = 0; i < 8; ++i) { u.hi[i] = (_Float16)0.0f; u.lo[i] = (_Float16)0.0f; } return u; }
};
// element e of a unit := v (rounded to the policy's operand type)
EO_DEV void set_elem(PBf16::U& u, int e, float v) { u[e] = (__bf16)v; }
EO_DEV void set_elem(PF32::U& u, int e, float v) { u[e] = v; }
EO_DEV void set_elem(PH3::U& u, int e, float v) { const _Float16 hv = (_Float16)v; u.hi[e] = hv; u.lo[e] = (_Float16)(v - (float)hv); }
EO_DEV float get_elem(const PBf16::U& u, int e) { return (float)u[e]; }
EO_DEV float get_elem(const PF32::U& u, int e) { return u[e]; }
EO_DEV float get_elem(const PH3::U& u, int e) { return (float)u.hi[e] + (float)u.lo[e]; }

// accumulator tile -> the KG32 B-operand units of the next layer (same feature order, no permutation)
template <class P> struct Units32 { typename P::U u[P::KG32]; };

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// two fp32 -> one packed bf16 word (round to nearest even): the pair form compiles to ONE v_cvt_pk_bf16_f32
EO_DEV uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2 p = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(p, bf16x2));
}
EO_DEV Units32<PBf16> pack_units(PBf16, const f32x16& v) {
    Units32<PBf16> o;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = cvt_pk_bf16(v[8 * s + 2 * i], v[8 * s + 2 * i + 1]);
        o.u[s] = __builtin_bit_cast(bf16x8, w);

    }
    return o;
}
EO_DEV Units32<PF32> pack_units(PF32, const f32x16& v) {
    Units32<PF32> o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) o.u[q][e] = v[4 * q + e];
    return o;
}

// ------------------------------------------------------------------------------------------------
// Weight stream: packed chunks (built by eonerf_pack.cpp) are double-buffered through LDS.
// chunk = G m-tiles; per m-tile KG x 1 KiB A units in fragment order, then G x 128 B of fp32 bias.
// ------------------------------------------------------------------------------------------------
struct ChunkDesc { uint32_t off, bytes; };
template <class P, int SLOT_BYTES> struct WStream {
    static_assert(SLOT_BYTES % 1024 == 0, "copy rounds are whole 1-KiB pieces");
    const uint8_t* g;            // packed stream (global)
    const ChunkDesc* tab;        // chunk table (global, read through the scalar cache)
    uint8_t* lds;                // 2 * SLOT_BYTES
    int n_chunks;
    int q;                       // index (in tab) of the chunk currently resident / being consumed
    uint32_t par;                // slot parity of the resident chunk
    int wave_b, lane_b;          // byte offsets of this wave's 1-KiB piece of a copy round (SGPR) and of the lane inside it
    bool late = false;           // wave stagger (run_layer): this wave runs a chunk's last epilogue behind the chunk barrier (wave-uniform)

    // descriptor of chunk qi.  Constant address space: the table is read-only for the whole launch, so this is an s_load
    // (a generic-pointer load becomes a VECTOR load + s_waitcnt vmcnt(0), which drains every outstanding slab store).
    EO_DEV ChunkDesc desc(int qi) const {
        const auto* t4 = reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(tab));
        return ChunkDesc{t4[2 * qi], t4[2 * qi + 1]};
    }
    // one ROUND of the copy of a chunk = one 16-byte LDS-DMA piece per thread (NT * 16 bytes)
    static constexpr uint32_t ROUND_B = P::NT * 16;
    static constexpr int MAX_ROUNDS = (SLOT_BYTES + ROUND_B - 1) / ROUND_B;
    uint32_t pf_off, pf_bytes, pf_base, pf_slot;     // prefetch in progress (wave-uniform)
    // One round = one 1-KiB LDS-DMA piece per wave.  Everything but the lane offset is wave-uniform (SGPR address, M0 from
    // scalar arithmetic, a scalar branch for the tail): a round costs the wave a handful of scalar instructions besides the
    // DMA itself.  The tail piece is copied whole (chunks are padded to 1 KiB by rounding the copy up: the slot is a multiple
    // of 1 KiB and the stream allocation carries 1 KiB of slack).
    // (buffer form: ONE descriptor for the launch, the piece enters through the scalar offset -- the flat form cost a 64-bit scalar add
    // and a 64-bit vector add per round, 240 rounds per sample tile)
    __amdgpu_buffer_rsrc_t rs;
    EO_DEV void round() {
        const uint32_t o = pf_base + wave_b;
        if (o < pf_bytes)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + pf_slot * SLOT_BYTES + o), 16,
                                                     lane_b, pf_off + o, 0, 0);
        pf_base += ROUND_B;
    }
    // first chunk of the launch
    EO_DEV void start() {
        q = 0; par = 0;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g), 0, -1, 0x00020000);
        const ChunkDesc d = desc(0);
        pf_off = d.off; pf_bytes = d.bytes; pf_base = 0; pf_slot = 0;
        while (pf_base < pf_bytes) round();
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // call before computing on the resident chunk: starts the prefetch of its successor into the other slot.  The copy
    // itself is issued by pump()/pump_rest() BETWEEN the MFMAs of the chunk's first m-tile (chunk_compute): issued in one
    // burst right after the barrier, every wave of the workgroup stalls on LDS-DMA issue at the same time with no matrix
    // work in flight.  All rounds are issued before the first m-tile's epilogue, i.e. before every slab store that
    // advance<> counts as younger.
    EO_DEV void prefetch_next() {
        int nq = q + 1; if (nq == n_chunks) nq = 0;
        const ChunkDesc d = desc(nq);
        pf_off = d.off; pf_bytes = (EO_ABL & 8) ? 0u : d.bytes; pf_base = 0; pf_slot = par ^ 1;
    }
    EO_DEV void pump() { if (pf_base < pf_bytes) round(); }
    EO_DEV void pump_rest() { while (pf_base < pf_bytes) round(); }
    // call after computing on the resident chunk.  YOUNGER = a lower bound on the vector-memory operations (the
    // epilogues' slab stores) this wave issued AFTER prefetch_next(): vmcnt retires in issue order, so waiting until
    // at most YOUNGER operations are outstanding guarantees the prefetch has landed without draining the stores
    // (a full vmcnt(0) here serialises every chunk behind an HBM write round trip).  Raw s_barrier: __syncthreads()
    // would add its own vmcnt(0).
#ifdef EO_STAMP
    unsigned long long t_wait = 0, t_bar = 0, t_flush = 0;
#endif
    template <int YOUNGER> EO_DEV void advance() {
        __builtin_amdgcn_sched_barrier(0);
#ifdef EO_STAMP
        const unsigned long long t0 = EO_T();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");
        const unsigned long long t1 = EO_T();
        asm volatile("s_barrier" ::: "memory");
        const unsigned long long t2 = EO_T();
        t_wait += t1 - t0; t_bar += t2 - t1;
        q = q + 1; if (q == n_chunks) q = 0;
        par ^= 1;
        return;
#endif
        if (EO_ABL & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");
        q = q + 1; if (q == n_chunks) q = 0;
        par ^= 1;
    }
    EO_DEV const uint8_t* cur() const { return lds + par * SLOT_BYTES; }
    // a wave WITHOUT samples in the current tile (balanced tail, TileSched): it still owns 1 KiB of every copy round and a seat at
    // every chunk barrier -- one pass over the chunk table, nothing else.  It has no younger stores to count: vmcnt(0).
    EO_DEV void idle_tile() {
        for (int i = 0; i < n_chunks; ++i) {
            prefetch_next();
            pump_rest();
            if (EO_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            q = q + 1; if (q == n_chunks) q = 0;
            par ^= 1;
        }
    }
};

// Tile schedule of the chain kernels.  The padded sample range is W = ceil(n_pts / TILE) * WAVES wave tiles of 32 samples.  Rounds in
// which every workgroup has a whole tile run tile = round * grid + block, as ever.  What is left (R < WAVES * grid wave tiles) used
// to be whole tiles for the first R / WAVES workgroups while the others idled through the last round: 7.49 rounds of work cost 8,
// 5.19 (the shadow pass) 6.  Now the R wave tiles are dealt out evenly, a contiguous run of floor / ceil (R / grid) per workgroup:
// every workgroup ends with ONE tile in which only its first `n` waves carry samples; the others step the weight stream
// (WStream::idle_tile).  A tile with half the waves costs well under a whole one (one wave per SIMD has the matrix pipe to itself).
template <class P> struct TileSched {
    int full, grid, block, rem_start, rem_cnt;
    EO_DEV TileSched(int n_pts, int grid_, int block_) : grid(grid_), block(block_) {
        const int W = (n_pts + P::TILE - 1) / P::TILE * P::WAVES;
        full = W / (P::WAVES * grid);
        const int R = W - full * P::WAVES * grid, q = R / grid, r = R % grid;
        rem_cnt = (EO_ABL & 32) ? 0 : q + (block < r ? 1 : 0);
        rem_start = full * P::WAVES * grid + block * q + (block < r ? block : r);
        if (EO_ABL & 32) {      // diagnostic: the former schedule (whole tiles only)
            const int tiles = R / P::WAVES;
            rem_cnt = block < tiles ? P::WAVES : 0;
            rem_start = (full * grid + block) * P::WAVES;
        }
    }
    EO_DEV int iters() const { return full + (rem_cnt > 0 ? 1 : 0); }
    // first wave tile and number of waves with samples in iteration `it`
    EO_DEV int first(int it) const { return it < full ? (it * grid + block) * P::WAVES : rem_start; }
    EO_DEV int waves(int it) const { return it < full ? P::WAVES : rem_cnt; }
};

template <class P> EO_DEV typename P::U lds_unit(const uint8_t* p) {
    if constexpr (P::IS_SPLIT) {      // hi and lo halves of the unit: two ds_read_b128
        typename P::U u;
        u.hi = *reinterpret_cast<const f16x8*>(p);
        u.lo = *reinterpret_cast<const f16x8*>(p + 1024);
        return u;
    } else {
        return *reinterpret_cast<const typename P::U*>(p);   // 16 B per lane -> ds_read_b128
    }
}

// acc tile initialised with the 32 bias values of its m-tile (bias region of the chunk, fp32)
EO_DEV f32x16 bias_init(const uint8_t* bias32, int h) {
    f32x16 acc;
    const float* b = reinterpret_cast<const float*>(bias32);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        f32x4 v = *reinterpret_cast<const f32x4*>(b + 8 * qd + 4 * h);
        acc[4 * qd + 0] = v[0]; acc[4 * qd + 1] = v[1]; acc[4 * qd + 2] = v[2]; acc[4 * qd + 3] = v[3];
    }
    return acc;
}
EO_DEV f32x16 zero_acc() { f32x16 a; for (int i = 0; i < 16; ++i) a[i] = 0.f; return a; }

// One chunk group of a layer:  for g in [0,G):  acc = bias + sum_kg A[g][kg] * B[kg];  epilogue of tile m0+g
// The A operands of the whole chunk form ONE stream of G*KG 1-KiB units; a rotating window of PF units is kept in
// flight ahead of the MFMA that consumes them (ds_read -> wait -> mfma per unit would expose the full LDS latency
// to every MFMA), across m-tile boundaries, and the next tile's bias is fetched while the current tile multiplies.
//
// Epilogues are SLICED and DEFERRED: epi(mt, acc, s) does slice s in [0, EPI_SLICES) of the epilogue of m-tile mt
// (accumulator registers 2s, 2s+1: convert, activation / mask, hand to the next layer, stage for the slab), and the
// slices of tile g run BETWEEN the MFMAs of tile g+1 of the same chunk -- a few VALU operations per MFMA gap instead of
// a block of ~50 behind the last MFMA of every tile (with the s_nop the matrix pipeline needs before its result can be
// read), during which this wave has no matrix work in flight.  The last tile of a chunk is finished before the chunk
// barrier.
// Chunk grouping of a layer (shared with the packer, eonerf_pack.cpp): its MT m-tiles are cut into the fewest chunks of at most
// CHUNK_KG_TARGET k-group units (and 8 m-tiles), as even as possible -- e.g. a 256 x 256 layer (KG = 16, MT = 8) into 3 + 3 + 2
// m-tiles.  Every chunk costs a workgroup barrier, so fewer, larger chunks; two chunk slots + the slab staging fill the LDS.
// (`target` = units a slot holds: 48 of 1 KiB for the bf16 / fp32 policies, whose kernels share the LDS with the slab staging; 32 of
//  2 KiB for the split policy, inference only)
constexpr int CHUNK_KG_TARGET = 48, CHUNK_KG_TARGET_SPLIT = 32;
template <class P> constexpr int chunk_target() { return P::IS_SPLIT ? CHUNK_KG_TARGET_SPLIT : CHUNK_KG_TARGET; }
__host__ __device__ constexpr int group_max(int kg, int mt, int target = CHUNK_KG_TARGET) {
    int g = target / kg;
    g = g > 8 ? 8 : g;
    g = g > mt ? mt : g;
    return g < 1 ? 1 : g;
}
__host__ __device__ constexpr int n_groups(int kg, int mt, int target = CHUNK_KG_TARGET) { return (mt + group_max(kg, mt, target) - 1) / group_max(kg, mt, target); }
__host__ __device__ constexpr int group_size(int kg, int mt, int c, int target = CHUNK_KG_TARGET) {
    return mt / n_groups(kg, mt, target) + (c < mt % n_groups(kg, mt, target) ? 1 : 0);
}
__host__ __device__ constexpr int group_start(int kg, int mt, int c, int target = CHUNK_KG_TARGET) {
    int s = 0;
    for (int k = 0; k < c; ++k) s += group_size(kg, mt, k, target);
    return s;
}
constexpr int EPI_SLICES = 8;
template <int KG> EO_DEV constexpr int slice_pos(int s) { return KG >= 14 ? 5 + s : (s * KG) / EPI_SLICES; }

// `last`: the accumulator of the chunk's LAST m-tile is handed back with its epilogue still to run (run_layer places it in front of or
// behind the chunk barrier, see there).
template <class P, int KG, int G, bool BIAS, class WS, class BArr, class Epi, class Mid>
EO_DEV void chunk_compute(WS& ws, int lane, int h, const BArr& B, int m0, Epi&& epi, Mid&& mid, f32x16& last) {
    const uint8_t* chunk = ws.cur();
    // prefetch rounds are issued after MFMAs 0, 1, 2, ... of the first m-tile (early: the copy has to land within the chunk);
    // the last position takes what is left
    constexpr int PLAST = WS::MAX_ROUNDS - 1 < KG ? WS::MAX_ROUNDS - 1 : KG - 1;
    constexpr int MIDK = KG > 4 ? 3 : KG - 1;        // where the oldest staged tile's slab flush is issued ...
    constexpr int MID0 = PLAST > MIDK ? PLAST : MIDK; // ... in the first m-tile: after the last prefetch round (the flush's stores
                                                      // must stay YOUNGER than the copy, see WStream::advance)
    constexpr int NF = G * KG;
    constexpr int PF = NF < 4 ? NF : (P::IS_SPLIT ? 3 : 4);      // (a split unit is two 16-byte reads and eight registers)
    const uint8_t* a = chunk + lane * 16;
    typename P::U fr[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) fr[d] = lds_unit<P>(a + d * P::UNIT_B);
    f32x16 acc = BIAS ? bias_init(chunk + NF * P::UNIT_B, h) : zero_acc();
    f32x16 pend = zero_acc();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        f32x16 nxt = zero_acc();
        // the next tile's bias is fetched late in this tile, once the pending tile's slices are done: its registers are then
        // free again (three live accumulator-sized sets would not fit 256 VGPRs)
        constexpr int BPOS = KG >= 15 ? 13 : KG - 1;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int f = g * KG + kg;
            if (BIAS && g + 1 < G && kg == BPOS) nxt = bias_init(chunk + NF * P::UNIT_B + (g + 1) * 128, h);
            acc = P::mma(fr[f % PF], B(kg), acc);
            if (!(EO_ABL & 2) && f + PF < NF) fr[f % PF] = lds_unit<P>(a + (f + PF) * P::UNIT_B);
            if (g == 0) {
                if (kg == PLAST) ws.pump_rest();
                else if (kg < PLAST) ws.pump();
            }
            // slab flush of the oldest staged tile in two phases: the transposing LDS reads here, the stores FDIST MFMAs later
            // behind a COUNTED lgkmcnt (only the A-unit reads issued in between may still be outstanding) -- a combined
            // read + s_waitcnt lgkmcnt(0) + store stalled the wave for a full LDS round trip once per m-tile
            {
                constexpr int FDIST = KG >= 8 ? 2 : 0;
                const int fk = g == 0 ? MID0 : MIDK;
                if (FDIST == 0) { if (kg == fk) mid.flush(); }
                else {
                    if (kg == fk) mid.flush_issue();
                    if (kg == fk + FDIST) {
                        // A-unit reads issued since (iterations fk+1 .. fk+FDIST): one each while the window still refills
                        const int f1 = g * KG + fk + 1, f2 = g * KG + fk + 2;
                        const int n_younger = ((f1 + PF < NF) ? 1 : 0) + ((f2 + PF < NF) ? 1 : 0);
                        if (n_younger == 2) mid.template flush_store<2>();
                        else if (n_younger == 1) mid.template flush_store<1>();
                        else mid.template flush_store<0>();
                    }
                }
            }
            if (g > 0) {
#pragma unroll
                for (int sl = 0; sl < EPI_SLICES; ++sl)
                    if (slice_pos<KG>(sl) == kg) epi(m0 + g - 1, pend, sl);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the window: the scheduler would otherwise re-serialise read/wait/mfma
        }
        if (g == G - 1) last = acc;
        else pend = acc;
        acc = nxt;
    }
}

// A whole layer: MT m-tiles in n_groups(KG, MT) chunks.  NST = slab stores every m-tile's epilogue issues (lower bound, 0 = unknown).
// `mid` provides the slab-flush hooks called a few MFMAs into every m-tile (or no-ops).
template <class P, int SLOT, int KG, int MT, bool BIAS, int NST = 0, int C = 0, class Mid, class BArr, class Epi>
EO_DEV void run_layer(WStream<P, SLOT>& ws, Mid&& mid, int lane, int h, const BArr& B, Epi&& epi) {
    constexpr int TG = chunk_target<P>();
    if constexpr (C < n_groups(KG, MT, TG)) {
        constexpr int G = group_size(KG, MT, C, TG), M0 = group_start(KG, MT, C, TG);
        static_assert(G * (KG * P::UNIT_B + 128) <= SLOT, "chunk does not fit the LDS slot");
        ws.prefetch_next();
        f32x16 last;
        chunk_compute<P, KG, G, BIAS>(ws, lane, h, B, M0, epi, mid, last);
        // The epilogue of the chunk's last m-tile has no MFMAs of its own wave to hide under.  WAVE STAGGER (ws.late, VERDICT r5 #2, cdna
        // guide "Two waves per SIMD" item 9): the two waves of a SIMD run the same program behind one barrier per chunk and tend to multiply
        // together and then run this epilogue together, the matrix pipe idle; a LATE wave (waves 4..7) runs it BEHIND the barrier instead,
        // beside its partner's first MFMAs of the next chunk, and finishes its own MFMAs while the partner is in its epilogue.  Same
        // arithmetic in the same order per wave: outputs bit for bit unchanged.
        // advance<>: stores younger than the copy (LOWER bound, see WStream::advance).  bf16: one flush (NST = 2 stores) per m-tile once
        // the staging queue runs: always from the layer's third tile on, and in every tile after the first of a later chunk (the flushes
        // sit between the MFMAs, in front of the barrier for every wave).  fp32 stores inside the slices: not counted.
        // (ONE copy of the epilogue in the code, the barrier on either side of it: a branch around the epilogue itself costs the kernel
        //  its register budget -- the operand arrays it writes would be live in two versions across the merge)
        if (ws.late) ws.template advance<(P::IS_BF16 ? (C == 0 ? (G >= 2 ? G - 2 : 0) : G - 1) * NST : 0)>();
#pragma unroll
        for (int sl = 0; sl < EPI_SLICES; ++sl) epi(M0 + G - 1, last, sl);
        if (!ws.late) ws.template advance<(P::IS_BF16 ? (C == 0 ? (G >= 2 ? G - 2 : 0) : G - 1) * NST : 0)>();
        run_layer<P, SLOT, KG, MT, BIAS, NST, C + 1>(ws, mid, lane, h, B, epi);
    }
}

// ------------------------------------------------------------------------------------------------
// ReLU + pack + 1-bit mask (forward), mask + pack (backward).  One mask dword covers a PAIR of m-tiles (32 regs/lane).
//   fp32: bit (16*(mt&1) + r) = acc[r] > 0.
//   bf16: works on the PACKED words (2 elements per VALU op): ReLU = v_pk_max_i16(w, 0) (a negative bf16 is a negative
//         int16), nonzero flags = v_pk_min_u16(w, 1), shifted into the mask with one v_lshl_or_b32 per word: word k
//         (elements 2k, 2k+1) of tile mt ends up at bit 8*(mt&1) + 7 - k, + 16 for the odd element.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) short s16x2;

// what one epilogue slice hands on: bf16 one packed word (elements 2s, 2s+1), fp32 the two values
template <class P> struct Sl;
template <> struct Sl<PBf16> { uint32_t w; };
template <> struct Sl<PF32> { float v0, v1; };
template <> struct Sl<PH3> { uint32_t hi, lo; };      // packed fp16 pairs

// forward: ReLU + mask bits of slice s; `bits` collects the tile's flags (reset by slice 0)
EO_DEV Sl<PBf16> relu_slice(PBf16, const f32x16& acc, int s, uint32_t& bits) {
    const s16x2 z = {0, 0};
    const uint32_t ones = 0x00010001u;
    const uint32_t wi = cvt_pk_bf16(acc[2 * s], acc[2 * s + 1]);
    if (EO_ABL & 4) return Sl<PBf16>{wi};
    const uint32_t x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wi), z));
    uint32_t t;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(x), "v"(ones));      // 1 where the element is > 0 (x is >= 0 here)
    bits = s == 0 ? t : ((bits << 1) | t);
    return Sl<PBf16>{x};
}
EO_DEV Sl<PF32> relu_slice(PF32, const f32x16& acc, int s, uint32_t& bits) {
    const float a0 = acc[2 * s], a1 = acc[2 * s + 1];
    const bool p0 = a0 > 0.f, p1 = a1 > 0.f;
    const uint32_t t = (p0 ? (1u << (2 * s)) : 0u) | (p1 ? (2u << (2 * s)) : 0u);
    bits = s == 0 ? t : (bits | t);
    return Sl<PF32>{p0 ? a0 : 0.f, p1 ? a1 : 0.f};
}
// split policy: two fp32 -> (hi, lo) packed fp16 pairs, round to nearest even
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
EO_DEV Sl<PH3> split_pair(float a0, float a1) {
    const f32x2 p = {a0, a1};
    const f16x2 hi = __builtin_convertvector(p, f16x2);
    const f32x2 back = __builtin_convertvector(hi, f32x2);
    const f32x2 rem = {a0 - back[0], a1 - back[1]};
    const f16x2 lo = __builtin_convertvector(rem, f16x2);
    return Sl<PH3>{__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo)};
}
// (no ReLU masks in this precision: inference only.  The `bits` / mask words carry the RANGE check instead: |v| > 65504 rounds to an
//  infinite hi half -- and lo = v - inf = -inf, a NaN operand that the next ReLU would silently turn into 0.  A running maximum of the
//  post-ReLU values of an m-tile in the slice word (one v_max3_f32 per slice), compared with 65504 once per m-tile (mask_commit) into a
//  wave-uniform flag that k_mlp_fwd looks at once per sample tile and eonerf_range_status reports.
//  A NaN accumulator must not vanish: `!(x <= 0)` keeps it through the ReLU (a plain x > 0 ? x : 0 turns it into 0), and the running maximum
//  is taken on the BIT patterns (post-ReLU values are >= 0, where unsigned order = float order; a NaN of either sign, sign bit masked, sorts
//  above +inf) -- fmaxf would drop it.)
EO_DEV Sl<PH3> relu_slice(PH3, const f32x16& acc, int s, uint32_t& bits) {
    const float a0 = !(acc[2 * s] <= 0.f) ? acc[2 * s] : 0.f, a1 = !(acc[2 * s + 1] <= 0.f) ? acc[2 * s + 1] : 0.f;
    const uint32_t b0 = __builtin_bit_cast(uint32_t, a0) & 0x7fffffffu, b1 = __builtin_bit_cast(uint32_t, a1) & 0x7fffffffu;
    const uint32_t m01 = b0 > b1 ? b0 : b1;
    bits = s == 0 ? m01 : (bits > m01 ? bits : m01);
    return split_pair(a0, a1);
}
EO_DEV Sl<PH3> relu_only_slice(PH3, const f32x16& acc, int s) { uint32_t b; return relu_slice(PH3(), acc, s, b); }
// once per m-tile: the tile's maximum against fp16's largest finite value, into a WAVE-UNIFORM word (scalar registers: the kernel has
// no vector register to spare for state that lives across a sample tile)
EO_DEV void mask_commit(PH3, int, uint32_t bits, uint32_t& m) {
    m |= __builtin_amdgcn_ballot_w64(bits > 0x477FE000u) != 0ull ? 1u : 0u;      // bits of 65504.f; inf and NaN patterns lie above
}
EO_DEV Sl<PH3> pack_slice(PH3, const f32x16& acc, int s) { return split_pair(acc[2 * s], acc[2 * s + 1]); }
EO_DEV void put_slice(PH3, F16Pair* arr, int mt, int s, const Sl<PH3>& v) {      // same place as the bf16 policy, in both halves
    u32x4 th = __builtin_bit_cast(u32x4, arr[2 * mt + (s >> 2)].hi), tl = __builtin_bit_cast(u32x4, arr[2 * mt + (s >> 2)].lo);
    th[s & 3] = v.hi; tl[s & 3] = v.lo;
    arr[2 * mt + (s >> 2)].hi = __builtin_bit_cast(f16x8, th);
    arr[2 * mt + (s >> 2)].lo = __builtin_bit_cast(f16x8, tl);
}
// forward, ReLU only (nobody reads this layer's mask bits)
EO_DEV Sl<PBf16> relu_only_slice(PBf16, const f32x16& acc, int s) {
    const s16x2 z = {0, 0};
    const uint32_t wi = cvt_pk_bf16(acc[2 * s], acc[2 * s + 1]);
    return Sl<PBf16>{__builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wi), z))};
}
EO_DEV Sl<PF32> relu_only_slice(PF32, const f32x16& acc, int s) {
    const float a0 = acc[2 * s], a1 = acc[2 * s + 1];
    return Sl<PF32>{a0 > 0.f ? a0 : 0.f, a1 > 0.f ? a1 : 0.f};
}
// after the last slice: fold the tile's flags into the mask dword of its tile PAIR
EO_DEV void mask_commit(PBf16, int mt, uint32_t bits, uint32_t& m) { m = (mt & 1) ? (m | (bits << 8)) : bits; }
EO_DEV void mask_commit(PF32, int mt, uint32_t bits, uint32_t& m) { m = (mt & 1) ? (m | (bits << 16)) : bits; }
// backward: slice s of (mask .* acc)
EO_DEV Sl<PBf16> mask_slice(PBf16, const f32x16& acc, int s, int mt, uint32_t m) {
    const int pos = 8 * (mt & 1) + 7 - s;
    const uint32_t sel = (m >> pos) & 0x00010001u;
    const uint32_t wi = cvt_pk_bf16(acc[2 * s], acc[2 * s + 1]);
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(wi), "v"(sel));      // x * {0,1} per element
    return Sl<PBf16>{r};
}
EO_DEV Sl<PF32> mask_slice(PF32, const f32x16& acc, int s, int mt, uint32_t m) {
    const uint32_t bits = m >> ((mt & 1) * 16 + 2 * s);
    const float a0 = acc[2 * s], a1 = acc[2 * s + 1];
    return Sl<PF32>{(bits & 1u) ? a0 : 0.f, (bits & 2u) ? a1 : 0.f};
}
// identity activation
EO_DEV Sl<PBf16> pack_slice(PBf16, const f32x16& acc, int s) { return Sl<PBf16>{cvt_pk_bf16(acc[2 * s], acc[2 * s + 1])}; }
EO_DEV Sl<PF32> pack_slice(PF32, const f32x16& acc, int s) { const float a0 = acc[2 * s], a1 = acc[2 * s + 1]; return Sl<PF32>{a0, a1}; }
// slice s of m-tile mt -> its place in the next layer's B-operand units (same feature order, no permutation)
EO_DEV void put_slice(PBf16, bf16x8* arr, int mt, int s, const Sl<PBf16>& v) {
    u32x4 t = __builtin_bit_cast(u32x4, arr[2 * mt + (s >> 2)]);
    t[s & 3] = v.w;
    arr[2 * mt + (s >> 2)] = __builtin_bit_cast(bf16x8, t);
}
EO_DEV void put_slice(PF32, f32x4* arr, int mt, int s, const Sl<PF32>& v) {
    arr[4 * mt + (s >> 1)][2 * (s & 1)] = v.v0;
    arr[4 * mt + (s >> 1)][2 * (s & 1) + 1] = v.v1;
}

// ------------------------------------------------------------------------------------------------
// Saved-activation / saved-gradient slabs: BLOCK-MAJOR tiled layout  [row block][sample tile][row in block][64 B]
//   one 64-byte segment = one feature row of one sample tile (bf16: 32 samples = the samples of one wave; fp32: 16).
//   A row block = the rows one weight-gradient job reads as an operand (a layer's 256 activations, the 64 encoding
//   slots, ...): block b = rows [s, s + r) of the global row numbering, and segment (row, tile t) lives at segment index
//   s * NT + t * r + (row - s)  (NT = sample tiles of the slab).  The weight-gradient GEMM therefore streams every operand
//   as ONE sequential run of r x 64 B chunks (measured +17 % HBM read rate over chunks 150 KB apart), and the eight waves
//   of a chain workgroup write eight adjacent chunks of the same block.
// ------------------------------------------------------------------------------------------------
constexpr int SEG_B = 64;
// cache policy of the slab stores: nt (streaming).  The slabs are written once and read once by a later kernel, far more data
// than the caches hold; with the default policy the write-back traffic held up the chain kernels (measured: forward chain
// -12 %, backward chain -21 %, and the weight-gradient GEMM that follows -5 %)
constexpr int SLAB_STORE_POLICY = 2;
#ifndef EO_ELEM_POLICY
#define EO_ELEM_POLICY 0      // cache policy of the element-wise slab stores (encoding rows, the heads' scalar gradient rows); nt (2) measured +0.3 % on the step: 2-byte streaming stores
#endif
struct SlabBlk { int s, r; };
template <class P> struct Slab {
    static constexpr int TSAMP = SEG_B / P::ACT_BYTES;     // samples per segment
    static constexpr int WAVE_TILES = 32 / TSAMP;          // sample tiles covered by one wave (bf16 1, fp32 2)
};
// Writer of one wave's share of a slab (Map::block(row) = the block of a row).  The wave holds a tile TRANSPOSED to what
// the slab wants (lane = sample, registers = features), and narrow stores are issue-bound on CDNA4 (~7 B/clk/CU for
// 8-byte stores), so
//   bf16: the packed tile goes through a wave-private LDS scratch ([32 samples][32 features], 72-B rows) and comes
//         back through ds_read_b64_tr_b16 (hardware transpose) as 8 consecutive samples of one feature per lane:
//         TWO buffer_store_dwordx4 per 32x32 tile instead of eight dword stores + DPP shuffles;
//   fp32: parity mode, plain dword stores (16 lanes = one 64-B segment per feature row).
constexpr int TR_STRIDE = 72;                    // scratch row stride: 18 dwords -> conflict-free ds_write_b64
constexpr int TR_WAVE_B = 32 * TR_STRIDE;        // 2304 B per wave
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <class P, class Map> struct SlabWriter;

// the `mid` argument of run_layer for kernels that save nothing
struct NoSlab {
    EO_DEV void flush() {}
    EO_DEV void flush_issue() {}
    template <int N> EO_DEV void flush_store() {}
};

struct SlabWriterBase {
    uint8_t* slab; uint32_t nt, tile0;           // wave-uniform: slab base, sample tiles of the slab, this wave's first sample tile
    EO_DEV __amdgpu_buffer_rsrc_t block_rs(SlabBlk b) const {      // descriptor of a whole block (SGPRs)
        return __builtin_amdgcn_make_buffer_rsrc(slab + (size_t)b.s * nt * SEG_B, 0, b.r * nt * SEG_B, 0x00020000);
    }
    EO_DEV uint32_t block_off(SlabBlk b, int row) const { return (((EO_ABL & 16) ? 0 : tile0) * b.r + (row - b.s)) * SEG_B; }   // EO_ABL 16: every wave writes sample tile 0 (stores stay in L2)
};

template <class Map> struct SlabWriter<PH3, Map> : NoSlab {      // never instantiated for training: the hooks of an inference chain
    static constexpr int LDS_BYTES = 0;
    static constexpr int FLUSH_STORES = 0;
    EO_DEV void init(void*, int, int, int, uint8_t*) {}
    EO_DEV void stage(int, int, const Sl<PH3>&) {}
    EO_DEV void elem(int, float) const {}
    EO_DEV void drain() {}
};

template <class Map> struct SlabWriter<PF32, Map> : SlabWriterBase {
    static constexpr int LDS_BYTES = 0;
    static constexpr int FLUSH_STORES = 0;          // tile() stores at once, flush_pending() is a no-op
    int c, h;
    EO_DEV void flush_pending() {}
    EO_DEV void flush() {}
    EO_DEV void flush_issue() {}
    template <int N> EO_DEV void flush_store() {}
    EO_DEV void drain() {}
    EO_DEV void init(void* slab_, int n_tiles, int wave_p0, int lane, uint8_t*) {
        c = lane & 31; h = lane >> 5;
        slab = reinterpret_cast<uint8_t*>(slab_); nt = n_tiles;
        tile0 = __builtin_amdgcn_readfirstlane(wave_p0) / 16;       // wave-uniform: keeps the descriptors in SGPRs
    }
    // lanes c >= 16 hold the samples of the wave's second sample tile: r x 64 B further on inside the block
    EO_DEV int voff(SlabBlk b) const { return (c >> 4) * b.r * SEG_B + 4 * h * SEG_B + (c & 15) * 4; }
    // slice s of the 32-row tile starting at row0: accumulator registers 2s, 2s+1
    EO_DEV void stage(int row0, int s, const Sl<PF32>& v) const {
        const SlabBlk b = Map::block(row0);
        const __amdgpu_buffer_rsrc_t rs = block_rs(b);
        const int vo = voff(b);
        const uint32_t so = block_off(b, row0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.v0), rs, vo, so + acc_row(2 * s, 0) * SEG_B, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.v1), rs, vo, so + acc_row(2 * s + 1, 0) * SEG_B, 0);
    }
    EO_DEV void elem(int row, float v) const {       // row (+4h through voff)
        const SlabBlk b = Map::block(row);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), block_rs(b), voff(b), block_off(b, row), 0);
    }
};

template <class Map> struct SlabWriter<PBf16, Map> : SlabWriterBase {
    // stage() writes the slices of a tile into a wave-private scratch buffer as they are produced (one ds_write_b64 per two
    // slices) and queues the tile when its last slice is in; flush_pending() -- called by the chain a few MFMAs into every
    // m-tile -- takes the OLDEST queued tile back through the transposing read and issues its two dwordx4 stores, so that the
    // LDS round trip (write -> transposed read -> store) is covered by matrix work.  Two scratch buffers, queue depth 2.
    static constexpr int LDS_BYTES = 8 * 2 * TR_WAVE_B;
    static constexpr int FLUSH_STORES = 2;
    __amdgpu_buffer_rsrc_t qa_rs, qb_rs; uint32_t qa_off, qb_off; int qa_buf, qb_buf;      // queued tiles: a = oldest (SGPRs)
    int n_pend = 0, buf = 0;
    uint32_t wprev;
    int voff1, svoff;
    uint8_t* wptr; const uint8_t* rptr;
    EO_DEV void init(void* slab_, int n_tiles, int wave_p0, int lane, uint8_t* scratch_wave) {
        const int c = lane & 31, h = lane >> 5;
        slab = reinterpret_cast<uint8_t*>(slab_); nt = n_tiles;
        tile0 = __builtin_amdgcn_readfirstlane(wave_p0) / 32;       // wave-uniform: keeps the descriptors in SGPRs (no waterfall loop)
        voff1 = 4 * h * SEG_B + c * 2;
        // scratch write: row = sample c, columns 8q + 4h .. +3 (natural feature order)
        n_pend = 0; buf = 0;
        wptr = scratch_wave + c * TR_STRIDE + h * 8;
        // transposed read: 16-lane group g = sample octet g; lane i = 4*qq + pp supplies (row 8g + 4t + qq, cols 16*pair +
        // 4pp..+3) and receives feature column 16*pair + i for those four samples.  Store `pair` therefore writes 16
        // features x 4 octets = 16 WHOLE 64-B segments (8 full 128-B lines) per instruction.
        const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
        rptr = scratch_wave + (8 * g + qq) * TR_STRIDE + pp * 8;
        svoff = i * SEG_B + g * 16;
    }
    // Inline asm on purpose: for the ds_read_tr intrinsic the compiler's wait-count pass assumes the read may alias the
    // in-flight LDS-DMA weight prefetch and puts s_waitcnt vmcnt(0) in front of it -- a full drain of the prefetch AND of
    // the previous tile's slab stores (an HBM write round trip) once per m-tile.  The scratch is wave-private and never
    // written by LDS-DMA.  The four result registers stay untouched between the two phases (tied through "+v").
    u32x2 fa0, fb0, fa1, fb1;
    int inflight = 0;
    EO_DEV void flush_issue() {
        if (!n_pend || inflight) return;
        const uint32_t ra = (uint32_t)(uintptr_t)(rptr + qa_buf * TR_WAVE_B);        // low 32 bits of a generic LDS address = LDS offset
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                     "ds_read_b64_tr_b16 %1, %4 offset:288\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:32\n\t"
                     "ds_read_b64_tr_b16 %3, %4 offset:320"
                     : "=&v"(fa0), "=&v"(fb0), "=&v"(fa1), "=&v"(fb1) : "v"(ra) : "memory");
        static_assert(4 * TR_STRIDE == 288, "asm offsets");
        inflight = 1;
    }
    // N = LDS operations issued after flush_issue() that may still be outstanding (LDS returns in order)
    template <int N> EO_DEV void flush_store() {
        if (!inflight) return;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa0), "+v"(fb0), "+v"(fa1), "+v"(fb1) : "n"(N) : "memory");
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{fa0[0], fa0[1], fb0[0], fb0[1]}, qa_rs, svoff, qa_off, SLAB_STORE_POLICY);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{fa1[0], fa1[1], fb1[0], fb1[1]}, qa_rs, svoff, qa_off + 16 * SEG_B, SLAB_STORE_POLICY);
        // A 128-bit store reads its data VGPRs a couple of cycles after issue.  The compiler only guards that window when the
        // store has no SGPR soffset; measured on gfx950 it exists with one too (a VALU write to the first data register right
        // behind the store reached memory instead of the tile: garbage in rows 28..31 of a 32x32 tile).
        asm volatile("s_nop 1" ::: "memory");
        qa_rs = qb_rs; qa_off = qb_off; qa_buf = qb_buf;
        --n_pend;
        inflight = 0;
    }
    EO_DEV void flush() { flush_issue(); flush_store<0>(); }
    EO_DEV void flush_pending() { flush(); }
    // slice s (one packed word = accumulator registers 2s, 2s+1) of the 32-row tile starting at row0
    EO_DEV void stage(int row0, int s, const Sl<PBf16>& v) {
        if (s == 0 && n_pend == 2) { flush_store<0>(); flush(); }   // both buffers queued: not in the regular schedule
        if (s & 1) *reinterpret_cast<u32x2*>(wptr + buf * TR_WAVE_B + 16 * (s >> 1)) = u32x2{wprev, v.w};
        else wprev = v.w;
        if (s == EPI_SLICES - 1) {
            const SlabBlk b = Map::block(row0);
            if (n_pend == 0) { qa_rs = block_rs(b); qa_off = block_off(b, row0); qa_buf = buf; }
            else { qb_rs = block_rs(b); qb_off = block_off(b, row0); qb_buf = buf; }
            ++n_pend;
            buf ^= 1;
        }
    }
    EO_DEV void drain() { flush_store<0>(); flush(); flush(); }
    EO_DEV void elem(int row, float v) const {
        const SlabBlk b = Map::block(row);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), block_rs(b), voff1, block_off(b, row), EO_ELEM_POLICY);
    }
};

// ------------------------------------------------------------------------------------------------
// Activation functions (PyTorch semantics: Softplus beta=1 threshold=20; Sigmoid)
// ------------------------------------------------------------------------------------------------
EO_DEV float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
EO_DEV float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// Positional-encoding slots (radiance_fields/mlp.py:190-208, L=10): 64 slots, 32 per lane half.
//   half 0, q<30 : sin(2^(q/3) * x[q%3])          q=30: x   q=31: y
//   half 1, q<30 : sin(2^(q/3) * x[q%3] + pi/2)   q=30: z   q=31: 0 (pad)
// q = kg*NE + e.  The packer maps slot -> reference encoding column with enc_col_of_slot().
// ------------------------------------------------------------------------------------------------
constexpr int ENC_SLOTS = 64;
__host__ __device__ constexpr int enc_col_of_hq(int h, int q) {
    return q < 30 ? (h ? 33 + q : 3 + q) : (q == 30 ? (h ? 2 : 0) : (h ? -1 : 1));
}

#define EO_PI_2_F 1.57079637050628662109375f   // fp32(0.5*math.pi), mlp.py:203
