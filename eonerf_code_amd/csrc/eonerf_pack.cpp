// Builds the flat parameter layout and the gather maps of the packed weight streams (see eonerf_pack.h).
// Tensor names / shapes follow the reference state_dict (radiance_fields/eonerf.py:84-139).
#include "eonerf_pack.h"
#include "eonerf_kernels.h"

void ParamLayout::build(int n_images) {
    n_img = n_images;
    t.clear();
    total = 0;
    auto add = [&](const std::string& name, int rows, int cols) {
        ParamInfo p{name, total, rows, cols};
        total += ((size_t)rows * cols + 3) & ~(size_t)3;     // 16-byte aligned tensors
        t.push_back(p);
        return (int)t.size() - 1;
    };
    // The flat buffer is also the layout of the gradient MESSAGE (one float per parameter + 4 control floats).  The EARLY block comes
    // first: the trunk layers whose gradients are complete when the camera pass' pipelined launch ends (layers 1-4, 6, 7: weights and
    // biases, 394,752 floats = 58 % of the message) -- a data-parallel trainer all-reduces [0, early) while the remaining gradient
    // kernels (GEMM jobs, tail) still run, and [early, end) behind them (trainer.py, eonerf_grad_early_floats).  Layer 5 is not in it:
    // its 63 skip columns come from the GEMM launch and interleave with the 256 pipelined ones row by row.  Tensors are found by
    // NAME (eonerf_param_info): nothing depends on their order in the buffer.
    const int trunk_in[8] = {63, 256, 256, 256, 256, 319, 256, 256};
    auto add_trunk = [&](int l) {
        trunk_w[l] = add("base_mlp.hidden_layers." + std::to_string(l) + ".weight", 256, trunk_in[l]);
        trunk_b[l] = add("base_mlp.hidden_layers." + std::to_string(l) + ".bias", 1, 256);
    };
    for (int l : {1, 2, 3, 4, 6, 7}) add_trunk(l);
    early = total;
    emb = add("transient_encoder.weight", n_images, 4);
    rad = add("radiometricT_enc.weight", n_images, 9);
    add_trunk(0);
    add_trunk(5);
    sig_w = add("sigma_layer.output_layer.weight", 1, 256);
    sig_b = add("sigma_layer.output_layer.bias", 1, 1);
    bot_w = add("bottleneck_layer.output_layer.weight", 256, 256);
    bot_b = add("bottleneck_layer.output_layer.bias", 1, 256);
    a1_w = add("albedo_mlp.hidden_layers.0.weight", 128, 256);
    a1_b = add("albedo_mlp.hidden_layers.0.bias", 1, 128);
    a2_w = add("albedo_mlp.output_layer.weight", 3, 128);
    a2_b = add("albedo_mlp.output_layer.bias", 1, 3);
    const int t_in[4] = {260, 128, 128, 128};
    for (int l = 0; l < 4; ++l) {
        t_w[l] = add("transient_mlp.hidden_layers." + std::to_string(l) + ".weight", 128, t_in[l]);
        t_b[l] = add("transient_mlp.hidden_layers." + std::to_string(l) + ".bias", 1, 128);
    }
    tsc_w = add("transient_scalar.output_layer.weight", 1, 128);
    tsc_b = add("transient_scalar.output_layer.bias", 1, 1);
    tbe_w = add("transient_beta.output_layer.weight", 1, 128);
    tbe_b = add("transient_beta.output_layer.bias", 1, 1);
    am1_w = add("ambient_mlp.hidden_layers.0.weight", 128, 27);
    am1_b = add("ambient_mlp.hidden_layers.0.bias", 1, 128);
    am2_w = add("ambient_mlp.output_layer.weight", 3, 128);
    am2_b = add("ambient_mlp.output_layer.bias", 1, 3);
}

static void slot_to_kghe(bool bf16, int slot, int& kg, int& h, int& e) {
    if (bf16) { kg = slot / 16; const int rem = slot % 16; e = (rem / 8) * 4 + rem % 4; h = (rem % 8) / 4; }
    else { kg = slot / 8; h = (slot % 8) / 4; e = slot % 4; }
}

int enc_col_of_slot(bool bf16, int slot) {
    int kg, h, e;
    slot_to_kghe(bf16, slot, kg, h, e);
    const int q = kg * (bf16 ? 8 : 4) + e;
    return enc_col_of_hq(h, q);
}

// prec: 0 fp32, 1 bf16, 2 the fp16 x 3 split (PH3: 16-feature k-groups in the bf16 fragment order, units of 2 KiB = [hi][lo])
void append_layer(PackedStream& s, int prec, const PackLayer& L) {
    const bool f32 = prec == 0, split = prec == 2;
    const int ne = f32 ? 4 : 8, esz = f32 ? 4 : 2, ub = split ? PH3::UNIT_B : 1024;
    const int tg = split ? CHUNK_KG_TARGET_SPLIT : CHUNK_KG_TARGET;
    for (int mg = 0; mg < n_groups(L.KG, L.MT, tg); ++mg) {          // the chunk grouping run_layer walks (eonerf_common.h)
        const int G = group_size(L.KG, L.MT, mg, tg), m0 = group_start(L.KG, L.MT, mg, tg);
        const uint32_t off = (uint32_t)s.bytes;
        const uint32_t bytes = (uint32_t)(G * (L.KG * ub + 128));
        s.chunks.push_back(ChunkDesc{off, bytes});
        for (int g = 0; g < G; ++g) {
            const int mt = m0 + g;
            for (int kg = 0; kg < L.KG; ++kg)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5;
                    for (int e = 0; e < ne; ++e) {
                        const int slot = f32 ? PF32::feat(kg, h, e) : PBf16::feat(kg, h, e);
                        PackEntry pe{off + (uint32_t)((g * L.KG + kg) * ub + lane * 16 + e * esz), L.w(32 * mt + r, slot)};
                        (f32 ? s.e32 : s.e16).push_back(pe);
                        if (split) { pe.dst += 1024; s.e16lo.push_back(pe); }
                    }
                }
            for (int i = 0; i < 32; ++i) {
                PackEntry pe{off + (uint32_t)(G * L.KG * ub + g * 128 + i * 4), L.bias ? L.b(32 * mt + i) : -1};
                s.e32.push_back(pe);
            }
        }
        s.bytes += bytes;
    }
}

PackedStream build_fwd_stream(const ParamLayout& pl, int prec, bool full) {
    PackedStream s;
    const bool bf16 = prec != 0;             // fragment order and k-group width of the 16-bit policies (bf16, fp16 x 3)
    const int KF = bf16 ? 16 : 8;
    const int HKG = 256 / KF, QKG = 128 / KF, EKG = 64 / KF;
    auto dense = [&](int wi, int bi, int out_rows, int in_cols, int KG, int MT) {
        PackLayer L{KG, MT, true,
            [=, &pl](int row, int slot) { return (row < out_rows && slot < in_cols) ? pl.at(wi, row, slot) : -1; },
            [=, &pl](int row) { return row < out_rows ? pl.at(bi, 0, row) : -1; }};
        append_layer(s, prec, L);
    };
    // trunk layer 0: encoding slots -> columns of W0
    append_layer(s, prec, PackLayer{EKG, 8, true,
        [&, bf16](int row, int slot) { const int c = enc_col_of_slot(bf16, slot); return c >= 0 ? pl.at(pl.trunk_w[0], row, c) : -1; },
        [&](int row) { return pl.at(pl.trunk_b[0], 0, row); }});
    for (int l = 1; l < 8; ++l) {
        if (l == 5) {   // [h(256), enc slots(64)] -> columns [0,256) and 256 + enc column (mlp.py:92-97)
            append_layer(s, prec, PackLayer{HKG + EKG, 8, true,
                [&, bf16](int row, int slot) {
                    if (slot < 256) return pl.at(pl.trunk_w[5], row, slot);
                    const int c = enc_col_of_slot(bf16, slot - 256);
                    return c >= 0 ? pl.at(pl.trunk_w[5], row, 256 + c) : -1;
                },
                [&](int row) { return pl.at(pl.trunk_b[5], 0, row); }});
        } else {
            dense(pl.trunk_w[l], pl.trunk_b[l], 256, 256, HKG, 8);
        }
    }
    if (!full) {      // one m-tile: the sigma row (sigma_layer)
        append_layer(s, prec, PackLayer{HKG, 1, true,
            [&](int row, int slot) { return row == 0 ? pl.at(pl.sig_w, 0, slot) : -1; },
            [&](int row) { return row == 0 ? pl.at(pl.sig_b, 0, 0) : -1; }});
        return s;
    }
    // m-tiles 0..3 = the albedo head's first layer FOLDED with the bottleneck layer (eonerf_pack.h: the kernels never evaluate the
    // bottleneck itself), m-tile 4 = sigma row.  (The sigma tile comes LAST: it saves nothing, and the counted wait of a chunk barrier
    // -- run_layer / WStream::advance -- relies on a slab flush from a layer's third m-tile on.)
    append_layer(s, prec, PackLayer{HKG, 5, true,
        [&](int row, int slot) { return row < 128 ? pl.fold_w(row, slot) : (row == 128 ? pl.at(pl.sig_w, 0, slot) : -1); },
        [&](int row) { return row < 128 ? pl.fold_b(row) : (row == 128 ? pl.at(pl.sig_b, 0, 0) : -1); }});
    dense(pl.a2_w, pl.a2_b, 3, 128, QKG, 1);
    // transient head's first layer on [X_8 (folded with the bottleneck), emb(img)]: slots 256..259 = embedding columns
    append_layer(s, prec, PackLayer{HKG + 1, 4, true,
        [&](int row, int slot) { return slot < 256 ? pl.fold_w(128 + row, slot) : (slot < 260 ? pl.at(pl.t_w[0], row, slot) : -1); },
        [&](int row) { return pl.fold_b(128 + row); }});
    for (int l = 1; l < 4; ++l) dense(pl.t_w[l], pl.t_b[l], 128, 128, QKG, 4);
    append_layer(s, prec, PackLayer{QKG, 1, true,
        [&](int row, int slot) { return row == 0 ? pl.at(pl.tsc_w, 0, slot) : (row == 1 ? pl.at(pl.tbe_w, 0, slot) : -1); },
        [&](int row) { return row == 0 ? pl.at(pl.tsc_b, 0, 0) : (row == 1 ? pl.at(pl.tbe_b, 0, 0) : -1); }});
    return s;
}

// Backward chain: dX^T = W^T dY^T.  "row" = INPUT feature of the forward layer, "slot" = OUTPUT feature.
PackedStream build_bwd_stream(const ParamLayout& pl, bool bf16, bool full, bool input_grad, bool transient, int heads) {
    PackedStream s;
    const int KF = bf16 ? 16 : 8;
    const int HKG = 256 / KF, QKG = 128 / KF;
    auto transposed = [&](int wi, int out_rows, int in_cols, int KG, int MT) {
        append_layer(s, bf16, PackLayer{KG, MT, false,
            [=, &pl](int row, int slot) { return (slot < out_rows && row < in_cols) ? pl.at(wi, slot, row) : -1; }, nullptr});
    };
    if (full) {
        if (transient) {
            // d{ts_pre, tb_pre} (slots 0,1) -> T4 (128)
            append_layer(s, bf16, PackLayer{1, 4, false,
                [&](int row, int slot) { return slot == 0 ? pl.at(pl.tsc_w, 0, row) : (slot == 1 ? pl.at(pl.tbe_w, 0, row) : -1); }, nullptr});
            for (int l = 3; l >= 1; --l) transposed(pl.t_w[l], 128, 128, QKG, 4);
        }
        // d albedo_pre (slots 0..2) -> A1 (128)
        transposed(pl.a2_w, 3, 128, 1, 4);
        // dY_T1 (slots 0..127) -> d embedding: rows 0..3 = the embedding columns 256..259 of the transient head's first layer
        if (transient)
            append_layer(s, bf16, PackLayer{QKG, 1, false,
                [&](int row, int slot) { return (row < 4 && slot < 128) ? pl.at(pl.t_w[0], slot, 256 + row) : -1; }, nullptr});
        // [dY_A1 (slots 0..127), dY_T1 (slots 128..255, with the transient head in the graph), d sigma_pre (next slot)] -> dX8, through the
        // FOLDED first layers of the heads (eonerf_pack.h): d bottleneck is never formed
        const int ns = transient ? 256 : 128;
        append_layer(s, bf16, PackLayer{ns / KF + 1, 8, false,
            [&, ns](int row, int slot) { return slot < ns ? pl.fold_w(slot, row) : (slot == ns ? pl.at(pl.sig_w, 0, row) : -1); }, nullptr});
    } else {
        append_layer(s, bf16, PackLayer{1, 8, false,
            [&](int row, int slot) { return slot == 0 ? pl.at(pl.sig_w, 0, row) : -1; }, nullptr});
    }
    if (heads == 1) return s;
    for (int l = 7; l >= 1; --l) {
        if (l == 5) {
            append_layer(s, bf16, PackLayer{HKG, input_grad ? 10 : 8, false,
                [&, bf16](int row, int slot) {
                    if (row < 256) return pl.at(pl.trunk_w[5], slot, row);
                    const int c = enc_col_of_slot(bf16, row - 256);
                    return c >= 0 ? pl.at(pl.trunk_w[5], slot, 256 + c) : -1;
                }, nullptr});
        } else {
            transposed(pl.trunk_w[l], 256, 256, HKG, 8);
        }
    }
    if (input_grad)
        append_layer(s, bf16, PackLayer{HKG, 2, false,
            [&, bf16](int row, int slot) { const int c = enc_col_of_slot(bf16, row); return c >= 0 ? pl.at(pl.trunk_w[0], slot, c) : -1; }, nullptr});
    return s;
}

// Layer-pipelined trunk backward: per stage (layer 7..1) the whole W_l^T as [m-tile 8][k-group 16] A units of 1 KiB
// (lane (r,h): input feature 32 mt + r, output features PBf16::feat(kg, h, e)) -- loaded ONCE into the stage's registers.
PackedStream build_pipe_stream(const ParamLayout& pl) {
    PackedStream s;
    for (int st = 0; st < PIPE_STAGES; ++st) {
        const int l = 7 - st;
        for (int mt = 0; mt < 8; ++mt)
            for (int kg = 0; kg < 16; ++kg)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5;
                    for (int e = 0; e < 8; ++e) {
                        const int o = PBf16::feat(kg, h, e);
                        s.e16.push_back(PackEntry{(uint32_t)(s.bytes + (size_t)(mt * 16 + kg) * 1024 + lane * 16 + e * 2), pl.at(pl.trunk_w[l], o, 32 * mt + r)});
                    }
                }
        s.bytes += 8 * 16 * 1024;
    }
    s.chunks.push_back(ChunkDesc{0, (uint32_t)s.bytes});
    return s;
}

// Input-gradient tail: rows = encoding slots (64 = 2 m-tiles), K = the 256 outputs of layer 0 (source 0) / layer 5 (source 1)
PackedStream build_ig_tail_stream(const ParamLayout& pl) {
    PackedStream s;
    for (int src = 0; src < 2; ++src)
        for (int mt = 0; mt < 2; ++mt)
            for (int kg = 0; kg < 16; ++kg)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5;
                    for (int e = 0; e < 8; ++e) {
                        const int o = PBf16::feat(kg, h, e);
                        const int col = enc_col_of_slot(true, 32 * mt + r);
                        const int idx = col < 0 ? -1 : (src == 0 ? pl.at(pl.trunk_w[0], o, col) : pl.at(pl.trunk_w[5], o, 256 + col));
                        s.e16.push_back(PackEntry{(uint32_t)(((size_t)((src * 2 + mt) * 16 + kg)) * 1024 + lane * 16 + e * 2), idx});
                    }
                }
    s.bytes = 2 * 2 * 16 * 1024;
    s.chunks.push_back(ChunkDesc{0, (uint32_t)s.bytes});
    return s;
}
