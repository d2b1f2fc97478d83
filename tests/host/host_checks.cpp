// Host-only checks of the C ABI's host logic, built for the CPU under the address + undefined-behaviour sanitizers (tests/test_host_sanitizers.py):
//   * eonerf_pack.cpp: the flat parameter layout and every packed weight stream (gather maps) for several image counts -- every
//     destination inside its stream, every source inside the flat buffer, no byte written twice, chunk tables consistent with the
//     grouping the chain kernels walk;
//   * eonerf_carve.h: the workspace layout for a set of (n_rays, flags, configuration) -- 256-byte alignment, no two buffers
//     overlapping, everything inside the reported size, measuring pass == carving pass.
// No HIP runtime call is made (GPU sanitizers are not available on this pool; the device side is covered by the parity tests).
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../eonerf_code_amd/csrc/eonerf_pack.h"
#include "../../eonerf_code_amd/csrc/eonerf_carve.h"

static int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #c); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static void check_stream(const char* name, const PackedStream& s, const ParamLayout& pl, bool chain) {
    std::vector<uint8_t> hit(s.bytes, 0);
    auto walk = [&](const std::vector<PackEntry>& e, int esz) {
        for (const PackEntry& pe : e) {
            CHECK((size_t)pe.dst + esz <= s.bytes, "%s: destination %u beyond the stream (%zu bytes)", name, pe.dst, s.bytes);
            // sources beyond the flat buffer index the fold buffer (ParamLayout::fold_w / fold_b, eonerf_pack.h)
            CHECK(pe.src >= -1 && (pe.src < 0 || (size_t)pe.src < pl.total + FOLD_FLOATS), "%s: source %d outside the flat + fold buffers (%zu + %d floats)", name, pe.src, pl.total, FOLD_FLOATS);
            CHECK(pe.dst % esz == 0, "%s: misaligned destination %u", name, pe.dst);
            if ((size_t)pe.dst + esz <= s.bytes)
                for (int k = 0; k < esz; ++k) { CHECK(!hit[pe.dst + k], "%s: byte %u written twice", name, pe.dst + k); hit[pe.dst + k] = 1; }
        }
    };
    walk(s.e16, 2);
    walk(s.e16lo, 2);
    walk(s.e32, 4);
    size_t covered = 0;
    for (uint8_t h : hit) covered += h;
    CHECK(covered == s.bytes, "%s: %zu of %zu bytes have a gather entry", name, covered, s.bytes);
    if (chain) {
        size_t off = 0;
        for (const ChunkDesc& c : s.chunks) {
            CHECK(c.off == off, "%s: chunk table not contiguous", name);
            const uint32_t slot = s.e16lo.empty() ? (uint32_t)(CHUNK_KG_TARGET * 1024 + 1024) : (uint32_t)(CHUNK_KG_TARGET_SPLIT * PH3::UNIT_B + 1024);
            CHECK(c.bytes <= slot, "%s: chunk of %u bytes exceeds the LDS slot", name, c.bytes);
            CHECK(c.bytes % 128 == 0, "%s: chunk size %u", name, c.bytes);
            off += c.bytes;
        }
        CHECK(off == s.bytes, "%s: chunks cover %zu of %zu bytes", name, off, s.bytes);
    }
}

static void check_layout(int n_img) {
    ParamLayout pl;
    pl.build(n_img);
    CHECK(pl.t.size() == 42, "parameter tensors: %zu", pl.t.size());      // the 44 state_dict entries minus the two int64 encoder buffers
    size_t end = 0;
    for (const ParamInfo& p : pl.t) {
        CHECK(p.offset % 4 == 0 && p.offset >= end, "%s overlaps its predecessor", p.name.c_str());
        end = p.offset + (size_t)p.rows * p.cols;
    }
    CHECK(end <= pl.total && pl.total - end < 4, "total %zu vs end %zu", pl.total, end);
    CHECK(pl.t[pl.trunk_w[5]].cols == 319 && pl.t[pl.t_w[0]].cols == 260 && pl.t[pl.emb].rows == n_img && pl.t[pl.rad].cols == 9, "shapes");
    for (int bf16 = 0; bf16 < 2; ++bf16) {
        const std::string tag = std::string(bf16 ? "bf16" : "fp32") + " n_img=" + std::to_string(n_img);
        check_stream(("fwd full " + tag).c_str(), build_fwd_stream(pl, bf16, true), pl, true);
        check_stream(("fwd dens " + tag).c_str(), build_fwd_stream(pl, bf16, false), pl, true);
        check_stream(("bwd full " + tag).c_str(), build_bwd_stream(pl, bf16, true, false), pl, true);
        check_stream(("bwd full ig " + tag).c_str(), build_bwd_stream(pl, bf16, true, true, true), pl, true);
        check_stream(("bwd rgb " + tag).c_str(), build_bwd_stream(pl, bf16, true, false, false), pl, true);
        check_stream(("bwd dens " + tag).c_str(), build_bwd_stream(pl, bf16, false, true), pl, true);
    }
    {   // fp16 x 3 split (inference): hi and lo halves of every unit, same sources
        const PackedStream sf = build_fwd_stream(pl, 2, true), sd = build_fwd_stream(pl, 2, false);
        check_stream("fwd full fp16x3", sf, pl, true);
        check_stream("fwd dens fp16x3", sd, pl, true);
        CHECK(sf.e16.size() == sf.e16lo.size() && sf.e16.size() == build_fwd_stream(pl, 1, true).e16.size(), "split stream: one hi and one lo entry per bf16-stream element");
        for (size_t k = 0; k < sf.e16.size(); ++k) CHECK(sf.e16lo[k].dst == sf.e16[k].dst + 1024 && sf.e16lo[k].src == sf.e16[k].src, "lo entry %zu", k);
    }
    check_stream("bwd full heads", build_bwd_stream(pl, true, true, false, true, 1), pl, true);
    check_stream("bwd rgb heads", build_bwd_stream(pl, true, true, false, false, 1), pl, true);
    {   // the fold: every element of the folded matrix and bias is a source of the full forward stream exactly once, and of the backward
        // streams (transient head in the graph) exactly once; the bottleneck layer's own weights are a source of NO chain stream
        const PackedStream ff = build_fwd_stream(pl, true, true), bf = build_bwd_stream(pl, true, true, false, true, 1), br = build_bwd_stream(pl, true, true, false, false, 1);
        std::vector<int> nf(FOLD_FLOATS, 0), nb(FOLD_FLOATS, 0), nr(FOLD_FLOATS, 0);
        const size_t bot_lo = pl.t[pl.bot_w].offset, bot_hi = pl.t[pl.bot_b].offset + 256;
        auto count = [&](const PackedStream& st, std::vector<int>& n) {
            for (const std::vector<PackEntry>* e : {&st.e16, &st.e32})
                for (const PackEntry& pe : *e) {
                    if (pe.src >= (int)pl.total) n[pe.src - pl.total]++;
                    CHECK(pe.src < 0 || (size_t)pe.src < bot_lo || (size_t)pe.src >= bot_hi, "bottleneck weight %d is a source of a chain stream", pe.src);
                }
        };
        count(ff, nf); count(bf, nb); count(br, nr);
        for (int k = 0; k < FOLD_FLOATS; ++k) {
            const bool bias = k >= 256 * 256, albedo = bias ? k - 256 * 256 < 128 : k < 128 * 256;
            CHECK(nf[k] == 1, "fold element %d: %d sources in the forward stream", k, nf[k]);
            CHECK(nb[k] == (bias ? 0 : 1), "fold element %d: %d sources in the backward stream", k, nb[k]);
            CHECK(nr[k] == ((bias || !albedo) ? 0 : 1), "fold element %d: %d sources in the rgb backward stream", k, nr[k]);
        }
    }
    check_stream("bwd dens heads", build_bwd_stream(pl, true, false, true, false, 1), pl, true);
    const PackedStream pw = build_pipe_stream(pl), iw = build_ig_tail_stream(pl);
    check_stream("pipe W^T", pw, pl, false);
    check_stream("ig tail W^T", iw, pl, false);
    CHECK(pw.bytes == (size_t)PIPE_STAGES * 8 * 16 * 1024, "pipe stream size");
    for (int s = 0; s < 64; ++s) {
        const int c16 = enc_col_of_slot(true, s), c32 = enc_col_of_slot(false, s);
        CHECK(c16 >= -1 && c16 < 63 && c32 >= -1 && c32 < 63, "encoding slot %d -> %d / %d", s, c16, c32);
    }
}

struct Span { const char* name; size_t lo, hi; };
static void add(std::vector<Span>& v, const char* name, const void* p, size_t bytes, const uint8_t* base) {
    if (p) v.push_back(Span{name, (size_t)(reinterpret_cast<const uint8_t*>(p) - base), (size_t)(reinterpret_cast<const uint8_t*>(p) - base) + bytes});
}
static void add_pass(std::vector<Span>& v, const PassBuffers& b, int n_rays, size_t p_cap, bool full, int ab, const uint8_t* base) {
    add(v, "counts", b.counts, 4 * (size_t)n_rays, base); add(v, "offsets", b.offsets, 4 * (size_t)(n_rays + 1), base); add(v, "n_pts", b.n_pts, 16, base);
    add(v, "px", b.px, 4 * p_cap, base); add(v, "py", b.py, 4 * p_cap, base); add(v, "pz", b.pz, 4 * p_cap, base);
    add(v, "tmid", b.tmid, 4 * p_cap, base); add(v, "delta", b.delta, 4 * p_cap, base); add(v, "simg", b.simg, 4 * p_cap, base);
    add(v, "sigma", b.sigma, 4 * p_cap, base); add(v, "albedo", b.albedo, 12 * p_cap, base); add(v, "ts", b.ts, 4 * p_cap, base); add(v, "tb", b.tb, 4 * p_cap, base);
    add(v, "act", b.act, (size_t)(full ? ACT_ROWS_FULL : ACT_ROWS_DENSITY) * p_cap * ab, base);
    add(v, "grd", b.grd, (size_t)(full ? GRD_ROWS_FULL : GRD_ROWS_DENSITY) * p_cap * ab, base);
    add(v, "masks", b.masks, (size_t)(full ? MASK_SLOTS_FULL : MASK_SLOTS_DENSITY) * p_cap * 32, base);
    add(v, "g_sigma", b.g_sigma, 4 * p_cap, base); add(v, "g_albedo", b.g_albedo, 12 * p_cap, base); add(v, "g_ts", b.g_ts, 4 * p_cap, base);
    add(v, "g_tb", b.g_tb, 4 * p_cap, base); add(v, "g_emb", b.g_emb, 16 * p_cap, base); add(v, "g_pos", b.g_pos, 12 * p_cap, base);
}

static void check_carve(const CarveCfg& cfg, int n_rays, int flags) {
    const RenderWs m = carve_render(cfg, nullptr, n_rays, flags);                 // measuring pass
    // a fake, never dereferenced base: only differences of pointers are formed
    uint8_t* base = reinterpret_cast<uint8_t*>((uintptr_t)1 << 40);
    const RenderWs w = carve_render(cfg, base, n_rays, flags);
    CHECK(w.bytes == m.bytes, "measuring pass %zu != carving pass %zu", m.bytes, w.bytes);
    const size_t p_cap = (size_t)p_cap_of(n_rays, cfg.n_samples);
    const bool od = flags & EONERF_F_ONLY_DEPTH;
    const int ab = cfg.bf16 ? 2 : 4;
    std::vector<Span> v;
    add(v, "cnt_first", w.cnt_first, 4 * (size_t)n_rays, base); add(v, "cnt_retry", w.cnt_retry, 4 * (size_t)n_rays, base); add(v, "flags", w.flags, 16, base);
    add(v, "ray_rec", w.ray_rec, 4 * (size_t)n_rays * RAY_REC, base); add(v, "g_ray", w.g_ray, 4 * (size_t)n_rays * RAY_REC, base);
    add(v, "amb_save", w.amb_save, 4 * (size_t)n_rays * 160, base);
    add(v, "m_bott+queue", w.m_bott, 4 * (BOTT_SCRATCH_F + 64), base);
    if (w.pipe.sync) {
        add(v, "sync", w.pipe.sync, PIPE_LAUNCHES * w.pipe.sync_bytes, base);
        add(v, "dy_in", w.pipe.dy_in, p_cap * 512, base);
        const size_t edges = (size_t)cfg.n_pipes * (PIPE_STAGES - 1);
        add(v, "rings", w.pipe.rings, edges * PIPE_RING * PIPE_UNIT_B, base);
        // the ONE memset of a backward call runs from m_bott to the end of the sync blocks: they must be adjacent
        CHECK(reinterpret_cast<uint8_t*>(w.pipe.sync) >= reinterpret_cast<uint8_t*>(w.m_bott) + 4 * (BOTT_SCRATCH_F + 64) &&
              reinterpret_cast<uint8_t*>(w.pipe.sync) - (reinterpret_cast<uint8_t*>(w.m_bott) + 4 * (BOTT_SCRATCH_F + 64)) < 256, "sync block not behind the GEMM queue");
        const size_t wgs = (size_t)cfg.n_pipes * PIPE_STAGES;
        CHECK(w.pipe.sync_bytes >= (64 + wgs * 32 + edges * 64) * 4, "sync block too small");
    }
    add(v, "pipe_part", w.det.pipe_part, (size_t)cfg.n_pipes * PIPE_STAGES * WGRAD_PART_F * 4, base);
    add(v, "wgrad_part", w.det.wgrad_part, (size_t)WGRAD_MAX_JOBS * 48 * WGRAD_PART_F * 4, base);
    add(v, "rad_rays", w.det.rad_rays, 24 * (size_t)n_rays, base); add(v, "emb_rays", w.det.emb_rays, 16 * (size_t)n_rays, base);
    add_pass(v, w.cam, n_rays, p_cap, !od, ab, base);
    add_pass(v, w.sun, n_rays, p_cap, false, ab, base);
    std::sort(v.begin(), v.end(), [](const Span& a, const Span& b) { return a.lo < b.lo; });
    for (size_t i = 0; i < v.size(); ++i) {
        CHECK(v[i].lo % 256 == 0, "%s not 256-byte aligned", v[i].name);
        CHECK(v[i].hi <= w.bytes, "%s ends at %zu beyond the workspace (%zu)", v[i].name, v[i].hi, w.bytes);
        if (i + 1 < v.size()) CHECK(v[i].hi <= v[i + 1].lo, "%s overlaps %s", v[i].name, v[i + 1].name);
    }
    const bool train = flags & EONERF_F_TRAIN;
    CHECK((w.cam.act != nullptr) == train && (w.g_ray != nullptr) == train, "training buffers");
    CHECK((w.sun.px != nullptr) == ((flags & EONERF_F_SHADOWS) && !od), "sun pass buffers");
}

int main() {
    for (int n_img : {1, 19, 20, 2048}) check_layout(n_img);
    CarveCfg cfgs[5];
    cfgs[0].bf16 = false;
    cfgs[1].pipe = true; cfgs[1].n_pipes = 36;
    cfgs[2] = cfgs[1]; cfgs[2].pipe_partials = true;
    cfgs[3] = cfgs[1]; cfgs[3].deterministic = cfgs[3].pipe_partials = true;
    cfgs[4] = cfgs[1]; cfgs[4].n_pipes = 1;
    const int flag_sets[] = {0, EONERF_F_SHADOWS, EONERF_F_ONLY_DEPTH, EONERF_F_TRAIN | EONERF_F_RGB_LOSS, EONERF_F_TRAIN | EONERF_F_SHADOWS,
                             EONERF_F_SHADOWS | EONERF_F_EVAL, EONERF_F_TRAIN | EONERF_F_SHADOWS | EONERF_F_EVAL};
    for (const CarveCfg& c : cfgs)
        for (int n_rays : {1, 37, 4096, 66050})
            for (int f : flag_sets) check_carve(c, n_rays, f);
    // the other step sizes (eonerf_set_n_samples): 64 and 256 samples per ray
    for (int ns : {64, 256}) {
        CarveCfg c = cfgs[1];
        c.n_samples = ns;
        for (int n_rays : {1, 37, 4096, 16384})
            for (int f : flag_sets) check_carve(c, n_rays, f);
    }
    CHECK(slab_blocks_addressable(true, (size_t)p_cap_of(66050, 128)) && !slab_blocks_addressable(true, (size_t)p_cap_of(66051, 128)), "bf16 size guard");
    CHECK(slab_blocks_addressable(false, (size_t)p_cap_of(33024, 128)) && !slab_blocks_addressable(false, (size_t)p_cap_of(33025, 128)), "fp32 size guard");
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("host checks ok\n");
    return 0;
}
