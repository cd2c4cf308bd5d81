// CVVP candidate scoring - the optional second ranking model of tts(cvvp_amount > 0) (reference: tortoise/models/cvvp.py:63-131,
// driven per conditioning clip at tortoise/api.py:464-472).  Two CollapsingTransformers (cvvp.py:19-51):
//   conditioning side: mel clip [80][T] -> Conv1d(k 5, stride 2) -> Conv1d(k 3, stride 2) -> tower -> to_conditioning_latent
//   speech side:       candidate codes -> embedding -> tower -> to_speech_latent
//   tower: the x-transformers Encoder CLVP also uses (xenc.h; ff_mult = 1) -> LayerNorm -> 1x1 conv -> AttentionBlock (GroupNorm32, 64-wide
//          heads, no relative positions) -> 1x1 conv -> mean over time
// score[b] = mean over clips of <normalize(cond latent), normalize(speech latent b)> * exp(temperature).  The conditioning latent of a clip
// does not depend on the candidate, so it is evaluated once per clip (the reference repeats the clip B times).  Nothing here is a new kernel:
// the stage is composed of the engine's GEMM / norm / flash / rotary launches in the token-major layout of the other stages; a stride-2
// convolution is the stride-1 tap GEMM followed by an even-row gather, as in cond.hip.
#include "xenc.h"

using namespace tt;

namespace {
__global__ void cvvp_even_rows_kernel(int* idx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = 2 * i;
}
}  // namespace

struct CvvpTower {
  tt_cvvp_tower w;
  std::vector<tt_clvp_layer> L;
};

struct tt_cvvp {
  tt_cvvp_config cfg;
  tt_cvvp_weights w;
  CvvpTower cond, speech;
  Arena arena;
  StreamBridge sb;
  size_t rows = 0;
  float* x = nullptr; void* h = nullptr; void* gg = nullptr; void* attn = nullptr;
  void* q = nullptr; void* k = nullptr; void* vt = nullptr;
  float* ha = nullptr; float* hb = nullptr;   // [rows][dim] f32 streams of the pre_combiner
  void* act = nullptr;                        // [rows][dim] T
  float* gn_partial = nullptr;
  float* pooled = nullptr; void* pooled_t = nullptr;
  float* cond_latent = nullptr;               // [16][dim]
  float* speech_latent = nullptr;             // [max sequences][dim]
  float* clip_scores = nullptr;               // [16][max sequences]
  float* mel_t = nullptr; void* mel_op = nullptr; void* c0 = nullptr; void* c0e = nullptr;  // conditioning front: [T][mel], [T][mel_pad], [T][dim/2], [T/2][dim/2]
  int* even_idx = nullptr;
  int max_seqs = 0;
  int* guard = nullptr;
  int* guard_host = nullptr;
};

// e->x holds the embedded rows [B * n][dim]; leaves the tower's latent rows in latent_out [B][dim]
static int cvvp_tower_run(tt_cvvp* e, const CvvpTower& t, int B, int n, float* latent_out, hipStream_t s) {
  const int D = e->cfg.dim, H = e->cfg.heads, dt = e->cfg.dtype;
  const int M = B * n, n_pad = round_up(n, 32);
  XencBufs xb{e->x, e->h, e->gg, e->attn, e->q, e->k, e->vt, e->guard};
  TT_TRY(xenc_layers_run(dt, xb, t.L.data(), e->cfg.depth, t.w.inv_freq, D, H, D, e->cfg.rot_dim, B, n, s));
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = e->x; a.ldx = D; a.M = M; a.D = D; a.mode = NORM_LAYER; a.g1 = t.w.norm_g; a.b1 = t.w.norm_b; a.eps1 = 1e-5f;
  a.out_t = e->h; a.ldot = D;
  a.guard = e->guard;
  TT_TRY(rownorm_launch(dt, a, s));
  // pre_combiner.0
  GemmArgs g = gemm_args(e->h, D, t.w.w_pre0, D, M, D, D);
  g.bias = t.w.b_pre0; g.out_f32 = e->ha; g.ldo32 = D;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  // pre_combiner.1: AttentionBlock (arch_util.py:80-123), one GroupNorm sample per sequence
  GroupNormArgs gn;
  memset(&gn, 0, sizeof(gn));
  gn.x = e->ha; gn.B = B; gn.S = n; gn.C = D; gn.gamma = t.w.attn.norm_g; gn.beta = t.w.attn.norm_b; gn.eps = 1e-5f; gn.act = ACT_NONE;
  gn.out_t = e->act; gn.ldot = D; gn.partial = e->gn_partial; gn.guard = e->guard;
  TT_TRY(groupnorm_launch(dt, gn, s));
  g = gemm_args(e->act, D, t.w.attn.w_qkv, D, M, 3 * D, D);
  g.bias = t.w.attn.b_qkv; g.seq_len = n; g.dmodel = D; g.heads = H; g.q = e->q; g.k = e->k; g.vt = e->vt; g.seq_pad = n_pad;
  g.q_scale = 0.125f;  // (q * 64^-1/4) . (k * 64^-1/4)
  TT_TRY(gemm_launch(dt, EPI_QKV_HEADS, g, s));
  FlashArgs f;
  memset(&f, 0, sizeof(f));
  f.q = e->q; f.k = e->k; f.vt = e->vt; f.out = e->attn; f.ldo = D; f.BH = B * H; f.heads = H; f.n = n; f.n_pad = n_pad;
  TT_TRY(flash_attention_launch(dt, f, s));
  g = gemm_args(e->attn, D, t.w.attn.w_proj, D, M, D, D);
  g.bias = t.w.attn.b_proj; g.res = e->ha; g.ldres = D; g.out_f32 = e->hb; g.ldo32 = D;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  // pre_combiner.2, then masked_mean with the all-ones eval mask (cvvp.py:46-51)
  TT_TRY(cast_pad_launch(dt, e->hb, D, e->act, D, M, D, D, s));
  g = gemm_args(e->act, D, t.w.w_pre2, D, M, D, D);
  g.bias = t.w.b_pre2; g.out_f32 = e->ha; g.ldo32 = D;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  TT_TRY(mean_rows_launch(e->ha, e->pooled, B, n, D, s));
  TT_TRY(cast_pad_launch(dt, e->pooled, D, e->pooled_t, D, B, D, D, s));
  g = gemm_args(e->pooled_t, D, t.w.w_latent, D, B, D, D);
  g.out_f32 = latent_out; g.ldo32 = D;
  return gemm_launch(dt, EPI_STD, g, s);
}

extern "C" {

int tt_cvvp_create(const tt_cvvp_config* cfg, const tt_cvvp_weights* w, tt_cvvp** out) {
  TT_REQUIRE(cfg && w && out, "tt_cvvp_create: null argument");
  TT_REQUIRE(cfg->dtype == DT_BF16 || cfg->dtype == DT_F16 || cfg->dtype == DT_F32, "tt_cvvp_create: unknown dtype %d", cfg->dtype);
  TT_REQUIRE(cfg->heads * 64 == cfg->dim && cfg->dim % 128 == 0 && cfg->depth >= 1, "tt_cvvp_create: unsupported dims (64-wide heads, dim a multiple of 128)");
  TT_REQUIRE(cfg->mel_pad % 64 == 0 && cfg->mel_pad >= cfg->mel_channels && cfg->max_rows >= 64 && cfg->max_cond_frames >= 32, "tt_cvvp_create: bad shape");
  TT_REQUIRE(w->speech_emb && w->temperature && w->w_cond0 && w->w_cond1 && w->cond.layers_host && w->speech.layers_host, "tt_cvvp_create: null weight");
  tt_cvvp* e = new tt_cvvp();
  e->cfg = *cfg;
  e->w = *w;
  e->cond.w = w->cond; e->cond.L.assign(w->cond.layers_host, w->cond.layers_host + cfg->depth);
  e->speech.w = w->speech; e->speech.L.assign(w->speech.layers_host, w->speech.layers_host + cfg->depth);
  const int D = cfg->dim;
  const size_t es = dtype_bytes(cfg->dtype);
  const size_t rows = (size_t)std::max(cfg->max_rows, cfg->max_cond_frames) + 64;
  e->rows = rows;
  e->max_seqs = cfg->max_rows / 8 + 8;  // a candidate has >= 8 codes
  int rc = e->sb.init();
  if (!rc) rc = e->arena.alloc_t(&e->x, rows * D);
  if (!rc) rc = e->arena.alloc(&e->h, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->gg, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->attn, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->q, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->k, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->vt, (size_t)D * (rows + 32 * (size_t)e->max_seqs) * es);
  if (!rc) rc = e->arena.alloc_t(&e->ha, rows * D);
  if (!rc) rc = e->arena.alloc_t(&e->hb, rows * D);
  if (!rc) rc = e->arena.alloc(&e->act, rows * D * es);
  if (!rc) rc = e->arena.alloc_t(&e->gn_partial, (rows / 16 + rows + 64) * 64);  // [samples][row chunks >= 16 rows][32][2], worst case as in diffusion.hip
  if (!rc) rc = e->arena.alloc_t(&e->pooled, (size_t)e->max_seqs * D);
  if (!rc) rc = e->arena.alloc(&e->pooled_t, (size_t)e->max_seqs * D * es);
  if (!rc) rc = e->arena.alloc_t(&e->cond_latent, (size_t)16 * D);
  if (!rc) rc = e->arena.alloc_t(&e->speech_latent, (size_t)e->max_seqs * D);
  if (!rc) rc = e->arena.alloc_t(&e->clip_scores, (size_t)16 * e->max_seqs);
  const size_t cf = (size_t)cfg->max_cond_frames + 64;
  if (!rc) rc = e->arena.alloc_t(&e->mel_t, cf * cfg->mel_pad);
  if (!rc) rc = e->arena.alloc(&e->mel_op, cf * cfg->mel_pad * es);
  if (!rc) rc = e->arena.alloc(&e->c0, cf * (D / 2) * es);
  if (!rc) rc = e->arena.alloc(&e->c0e, cf * (D / 2) * es);
  if (!rc) rc = e->arena.alloc_t(&e->even_idx, cf);
  if (!rc) rc = e->arena.alloc_t(&e->guard, 4);
  if (!rc && hipHostMalloc((void**)&e->guard_host, 4 * sizeof(int)) != hipSuccess) { set_error("tt_cvvp_create: hipHostMalloc failed"); rc = -2; }
  if (!rc) {
    e->guard_host[0] = 0;
    cvvp_even_rows_kernel<<<cdiv((int)cf, 256), 256>>>(e->even_idx, (int)cf);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_error("tt_cvvp_create: index fill failed"); rc = -2; }
  }
  if (rc) {
    tt_cvvp_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_cvvp_destroy(tt_cvvp* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  if (e->guard_host) (void)hipHostFree(e->guard_host);
  e->arena.release();
  e->sb.destroy();
  delete e;
}

int tt_cvvp_score(tt_cvvp* e, const float* mels, int n_clips, int T, const int* codes, int B, int n, float* scores, void* stream) {
  TT_REQUIRE(e && mels && codes && scores, "tt_cvvp_score: null argument");
  TT_REQUIRE(n_clips >= 1 && n_clips <= 16 && T >= 29 && T <= e->cfg.max_cond_frames, "tt_cvvp_score: %d clips of %d frames (1 .. 16 clips, 29 .. %d frames)", n_clips, T, e->cfg.max_cond_frames);
  TT_REQUIRE(B >= 1 && n >= 8 && (size_t)B * n <= (size_t)e->cfg.max_rows && B <= e->max_seqs, "tt_cvvp_score: B=%d n=%d exceed capacity %d rows (n must be >= 8)", B, n, e->cfg.max_rows);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int D = e->cfg.dim, D2 = D / 2, MC = e->cfg.mel_channels, MP = e->cfg.mel_pad, dt = e->cfg.dtype;
  const int es = dtype_bytes(dt);
  const int T2 = (T + 1) / 2, T3 = (T2 + 1) / 2;  // Conv1d(stride 2) with "same" padding: ceil(n / 2) outputs, output j = the stride-1 result at 2 j
  for (int c = 0; c < n_clips; ++c) {
    TT_TRY(transpose_launch(mels + (size_t)c * MC * T, e->mel_t, MC, T, s));          // [mel][T] -> [T][mel]
    TT_TRY(cast_pad_launch(dt, e->mel_t, MC, e->mel_op, MP, T, MC, MP, s));
    GemmArgs g = gemm_args(e->mel_op, MP, e->w.w_cond0, 5 * MP, T, D2, 5 * MP);
    g.taps = 5; g.seq_len = T; g.bias = e->w.b_cond0; g.out_t = e->c0; g.ldot = D2;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    TT_TRY(gather_rows_launch((const float*)e->c0, e->even_idx, (float*)e->c0e, T2, D2 * es / 4, s));  // rows of D2 elements as 4-byte words
    g = gemm_args(e->c0e, D2, e->w.w_cond1, 3 * D2, T2, D, 3 * D2);
    g.taps = 3; g.seq_len = T2; g.bias = e->w.b_cond1; g.out_f32 = e->ha; g.ldo32 = D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    TT_TRY(gather_rows_launch(e->ha, e->even_idx, e->x, T3, D, s));
    TT_TRY(cvvp_tower_run(e, e->cond, 1, T3, e->cond_latent + (size_t)c * D, s));
  }
  TT_TRY(gather_rows_launch(e->w.speech_emb, codes, e->x, B * n, D, s));
  TT_TRY(cvvp_tower_run(e, e->speech, B, n, e->speech_latent, s));
  for (int c = 0; c < n_clips; ++c)
    TT_TRY(clvp_score_launch(e->cond_latent + (size_t)c * D, 1, e->speech_latent, e->w.temperature, e->clip_scores + (size_t)c * B, B, D, s));
  TT_TRY(mean_rows_launch(e->clip_scores, scores, 1, n_clips, B, s));                   // mean over the clips (api.py:468)
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s));
  return e->sb.leave(us);
}

int tt_cvvp_guard(tt_cvvp* e, int reset) {
  if (!e) { set_error("tt_cvvp_guard: null handle"); return -1; }
  const int n = e->guard_host[0];
  if (n > 0) set_error("CVVP stage: %d kernel(s) met non-finite values (operand overflow in %s)", n, e->cfg.dtype == DT_F16 ? "fp16: use bf16 operands for this stage" : "bf16");
  if (reset && n > 0) {
    if (hipMemsetAsync(e->guard, 0, 4 * sizeof(int), e->sb.own) != hipSuccess || hipStreamSynchronize(e->sb.own) != hipSuccess) { set_error("tt_cvvp_guard: reset failed"); return -2; }
    e->guard_host[0] = 0;
  }
  return n;
}

}  // extern "C"
