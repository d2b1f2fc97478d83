// Host-side description of the flat parameter buffer and of the packed weight streams.
//
// A stream is the concatenation of chunks in the exact order a chain kernel consumes them (eonerf_mlp_fwd.hip,
// eonerf_mlp_bwd.hip).  chunk = G m-tiles:  [G][KG] units of 1 KiB (A operand in MFMA fragment order: lane (r,h)
// holds row 32*m + r, features P::feat(kg, h, e)), then G x 32 fp32 biases.  The packer is a pure gather:
// every destination element has a source index into the flat fp32 parameter buffer (or -1 = zero).
#pragma once
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include "eonerf_common.h"

struct ParamInfo { std::string name; size_t offset; int rows, cols; };

struct ParamLayout {
    std::vector<ParamInfo> t;
    size_t total = 0;
    size_t early = 0;      // [0, early): the tensors whose gradients the pipelined camera launch completes (see build)
    int n_img = 0;
    // indices into t
    int emb, rad, trunk_w[8], trunk_b[8], sig_w, sig_b, bot_w, bot_b, a1_w, a1_b, a2_w, a2_b;
    int t_w[4], t_b[4], tsc_w, tsc_b, tbe_w, tbe_b, am1_w, am1_b, am2_w, am2_b;
    void build(int n_images);
    int at(int ti, int r, int c) const { return (int)(t[ti].offset + (size_t)r * t[ti].cols + c); }
    // FOLDED head weights (see FOLD_FLOATS): gather sources beyond the flat parameter buffer -- index total + k = element k of the
    // context's fold buffer.  Row o < 128: albedo head's first layer, o >= 128: transient head's first layer (its 256 bottleneck columns)
    int fold_w(int o, int i) const { return (int)(total + (size_t)o * 256 + i); }
    int fold_b(int o) const { return (int)(total + 256 * 256 + o); }
};
// The bottleneck layer has an IDENTITY activation (radiance_fields/eonerf.py:108-113: output_activation = nn.Identity), so the first
// layers of the two heads that read it compose with it exactly:
//     ReLU(W_A1 (W_b x + b_b) + b_A1) = ReLU((W_A1 W_b) x + (W_A1 b_b + b_A1))          (and W_T1[:, :256] likewise)
// The chain kernels therefore never evaluate the bottleneck layer: they multiply X_8 by the folded 256 x 256 matrix [W_A1 W_b; W_T1' W_b]
// (fp32 product of the fp32 master weights, re-computed by k_fold after every optimizer step, rounded ONCE to the stream's precision),
// forward and backward -- 65,536 MACs per camera sample fewer in each direction.  The three layers' weight gradients follow from the
// bottleneck factors M = [dA1; dT1]^T X_8 as before (BottWgradArgs).  Layout of the fold buffer: [256][256] W_f | [256] b_f.
constexpr int FOLD_FLOATS = 256 * 256 + 256;

struct PackEntry { uint32_t dst; int32_t src; };     // dst: byte offset into the stream

struct PackedStream {
    std::vector<ChunkDesc> chunks;
    std::vector<PackEntry> e16;     // 16-bit destinations (bf16; fp16 x 3 split: the hi halves)
    std::vector<PackEntry> e16lo;   // fp16 x 3 split: the lo halves (value - float(fp16(value)), rounded to fp16)
    std::vector<PackEntry> e32;     // fp32 destinations
    size_t bytes = 0;
};

struct PackLayer {
    int KG, MT;
    bool bias;
    std::function<int(int row, int slot)> w;     // flat index of the weight multiplying input-slot `slot` for output row `row`
    std::function<int(int row)> b;               // flat index of the bias of output row `row`
};

void append_layer(PackedStream& s, int prec, const PackLayer& L);      // prec: 0 fp32, 1 bf16, 2 fp16 x 3 split (forward streams only)
PackedStream build_fwd_stream(const ParamLayout& pl, int prec, bool full);
// heads: 0 = the whole chain; 1 = the stream ends behind the [dY_A1, dY_T1, d sigma_pre] -> dX8 layer (the chain kernel's PIPE 1 variant
// leaves the trunk to eonerf_bwd_pipe.hip).  The stream is consumed cyclically, so it must hold exactly the layers one tile walks.
PackedStream build_bwd_stream(const ParamLayout& pl, bool bf16, bool full, bool input_grad, bool transient = true, int heads = 0);
PackedStream build_pipe_stream(const ParamLayout& pl);      // stage-stationary W_l^T of trunk layers 7..1 (eonerf_bwd_pipe.hip), bf16
PackedStream build_ig_tail_stream(const ParamLayout& pl);   // W_0^T and the skip columns of W_5^T as A units (eonerf_ig_tail.hip), bf16
int enc_col_of_slot(bool bf16, int slot);      // reference encoding column (mlp.py:190-208) of an encoding slot, -1 = pad
