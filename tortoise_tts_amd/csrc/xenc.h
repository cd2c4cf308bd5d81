// The x-transformers Encoder stack that CLVP's towers (clvp.hip) and CVVP's CollapsingTransformers (cvvp.hip) share
// (reference: tortoise/models/xtransformers.py:731-1013 as built at clvp.py:54-83 / cvvp.py:23-35): per layer pre-RMSNorm ->
// bias-free q / k / v -> rotary on the first rot_dim dims of q, k AND v -> softmax(q k^T / 8) v -> to_out + residual; pre-RMSNorm ->
// GEGLU feed-forward (value * gelu(gate) formed in the projection's epilogue) -> residual.  Token-major f32 residual stream.
#pragma once
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"

namespace tt {

struct XencBufs {
  float* x;    // [M][D] f32 residual stream (in / out)
  void* h;     // [M][D] T normalised rows
  void* gg;    // [M][inner] T GEGLU output
  void* attn;  // [M][D] T attention output
  void* q;     // flash layouts
  void* k;
  void* vt;
  int* guard;  // operand-overflow counter of the stage
};

static inline int xenc_layers_run(int dt, const XencBufs& b, const tt_clvp_layer* layers, int depth, const float* inv_freq, int D, int H, int inner,
                                  int rot_dim, int B, int n, hipStream_t s) {
  const int M = B * n, n_pad = round_up(n, 32);
  for (int l = 0; l < depth; ++l) {
    const tt_clvp_layer& w = layers[l];
    RowNormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = b.x; a.ldx = D; a.M = M; a.D = D; a.mode = NORM_RMS; a.g1 = w.attn_norm_g; a.eps1 = 1e-8f;
    a.out_t = b.h; a.ldot = D;
    a.guard = b.guard;
    TT_TRY(rownorm_launch(dt, a, s));
    GemmArgs g = gemm_args(b.h, D, w.w_qkv, D, M, 3 * D, D);
    g.seq_len = n; g.dmodel = D; g.heads = H; g.q = b.q; g.k = b.k; g.vt = b.vt; g.seq_pad = n_pad; g.q_scale = 0.125f;
    TT_TRY(gemm_launch(dt, EPI_QKV_HEADS, g, s));
    TT_TRY(rotary_launch(dt, b.q, b.k, b.vt, inv_freq, B * H, n, n_pad, rot_dim, s));
    FlashArgs f;
    memset(&f, 0, sizeof(f));
    f.q = b.q; f.k = b.k; f.vt = b.vt; f.out = b.attn; f.ldo = D; f.BH = B * H; f.heads = H; f.n = n; f.n_pad = n_pad;
    TT_TRY(flash_attention_launch(dt, f, s));
    g = gemm_args(b.attn, D, w.w_out, D, M, D, D);
    g.bias = w.b_out; g.res = b.x; g.ldres = D; g.out_f32 = b.x; g.ldo32 = D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    a.g1 = w.ff_norm_g;
    TT_TRY(rownorm_launch(dt, a, s));
    // GEGLU (xtransformers.py:429-437): value * gelu(gate) formed in the projection's epilogue (value / gate rows interleaved at pack
    // time) - the [M][2 inner] projection (315 MB at 256 candidates x 200 codes of CLVP) is never written or re-read
    g = gemm_args(b.h, D, w.w_ff1, D, M, 2 * inner, D);
    g.bias = w.b_ff1; g.out_t = b.gg; g.ldot = inner;
    TT_TRY(gemm_launch(dt, EPI_GEGLU, g, s));
    g = gemm_args(b.gg, inner, w.w_ff2, inner, M, D, inner);
    g.bias = w.b_ff2; g.res = b.x; g.ldres = D; g.out_f32 = b.x; g.ldo32 = D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  }
  return 0;
}

}  // namespace tt
