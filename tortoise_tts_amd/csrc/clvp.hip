// CLVP candidate scoring (reference: tortoise/models/clvp.py:99-135 over the x-transformers Encoder,
// tortoise/models/xtransformers.py:731-1013): pre-RMSNorm, bias-free q/k/v, rotary on the first 32
// dims of q, k and v, softmax(q k^T / 8), GEGLU feed-forward, final LayerNorm, mean pool, latent
// projection, cosine similarity * exp(temperature).  The text tower runs once per utterance (the
// reference repeats the prompt B times, api.py:463); the speech tower runs over B x n code rows.
#include "xenc.h"

using namespace tt;

struct ClvpTower {
  tt_clvp_tower w;
  std::vector<tt_clvp_layer> L;
};

struct tt_clvp {
  tt_clvp_config cfg;
  ClvpTower text, speech;
  const float* temperature;
  Arena arena;
  StreamBridge sb;
  float* x = nullptr; void* h = nullptr; void* gg = nullptr; void* attn = nullptr;
  void* q = nullptr; void* k = nullptr; void* vt = nullptr;
  float* enc = nullptr; float* pooled = nullptr; void* pooled_t = nullptr;
  float* text_latent = nullptr; float* speech_latent = nullptr;
  int max_batch = 0;
  int* guard = nullptr;       // operand-overflow guard (see tt_ar_guard): bumped by the row norms
  int* guard_host = nullptr;  // pinned copy, refreshed at the end of tt_clvp_score
};

static int clvp_tower_run(tt_clvp* e, const ClvpTower& t, const int* tokens, int B, int n, float* latent_out, hipStream_t s) {
  const int D = e->cfg.dim, H = e->cfg.heads, inner = e->cfg.ff_inner, dt = e->cfg.dtype;
  const int M = B * n;
  TT_TRY(gather_rows_launch(t.w.emb, tokens, e->x, M, D, s));
  XencBufs xb{e->x, e->h, e->gg, e->attn, e->q, e->k, e->vt, e->guard};
  TT_TRY(xenc_layers_run(dt, xb, t.L.data(), e->cfg.depth, t.w.inv_freq, D, H, inner, e->cfg.rot_dim, B, n, s));
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = e->x; a.ldx = D; a.M = M; a.D = D; a.mode = NORM_LAYER; a.g1 = t.w.norm_g; a.b1 = t.w.norm_b; a.eps1 = 1e-5f;
  a.out_f32 = e->enc; a.ldo32 = D;
  a.guard = e->guard;
  TT_TRY(rownorm_launch(dt, a, s));
  TT_TRY(mean_rows_launch(e->enc, e->pooled, B, n, D, s));
  TT_TRY(cast_pad_launch(dt, e->pooled, D, e->pooled_t, D, B, D, D, s));
  GemmArgs g = gemm_args(e->pooled_t, D, t.w.w_latent, D, B, e->cfg.latent_dim, D);
  g.out_f32 = latent_out; g.ldo32 = e->cfg.latent_dim;
  return gemm_launch(dt, EPI_STD, g, s);
}

extern "C" {

int tt_clvp_create(const tt_clvp_config* cfg, const tt_clvp_tower* text, const tt_clvp_tower* speech, const float* temperature,
                   tt_clvp** out) {
  TT_REQUIRE(cfg && text && speech && temperature && out, "tt_clvp_create: null argument");
  TT_REQUIRE(cfg->heads * 64 == cfg->dim && cfg->dim % 64 == 0 && cfg->ff_inner % 64 == 0, "tt_clvp_create: unsupported dims");
  TT_REQUIRE(cfg->dtype == DT_BF16 || cfg->dtype == DT_F16 || cfg->dtype == DT_F32, "tt_clvp_create: unknown dtype %d", cfg->dtype);
  tt_clvp* e = new tt_clvp();
  e->cfg = *cfg;
  e->text.w = *text; e->text.L.assign(text->layers_host, text->layers_host + cfg->depth);
  e->speech.w = *speech; e->speech.L.assign(speech->layers_host, speech->layers_host + cfg->depth);
  e->temperature = temperature;
  const size_t rows = (size_t)cfg->max_rows + 64;
  const int D = cfg->dim;
  const size_t es = dtype_bytes(cfg->dtype);
  e->max_batch = cfg->max_rows;  // every row could be its own sequence in the worst case
  int rc = e->sb.init();
  if (!rc) rc = e->arena.alloc_t(&e->x, rows * D);
  if (!rc) rc = e->arena.alloc(&e->h, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->gg, rows * cfg->ff_inner * es);
  if (!rc) rc = e->arena.alloc(&e->attn, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->q, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->k, rows * D * es);
  // V^T: [B*H][64][n_pad]; n_pad <= n + 31 and B*n <= max_rows, B <= max_rows / 1
  if (!rc) rc = e->arena.alloc(&e->vt, (size_t)D * (rows + 32 * (size_t)(cfg->max_rows / 8 + 8)) * es);
  if (!rc) rc = e->arena.alloc_t(&e->enc, rows * D);
  if (!rc) rc = e->arena.alloc_t(&e->pooled, rows * D / 8 + D);
  if (!rc) rc = e->arena.alloc(&e->pooled_t, (rows * D / 8 + D) * es);
  if (!rc) rc = e->arena.alloc_t(&e->text_latent, 16 * (size_t)cfg->latent_dim);  // one row per utterance of a tt_clvp_score_groups call
  if (!rc) rc = e->arena.alloc_t(&e->speech_latent, (rows / 8 + 8) * cfg->latent_dim);
  if (!rc) rc = e->arena.alloc_t(&e->guard, 4);
  if (!rc && hipHostMalloc((void**)&e->guard_host, 4 * sizeof(int)) != hipSuccess) { set_error("tt_clvp_create: hipHostMalloc failed"); rc = -2; }
  if (!rc) e->guard_host[0] = 0;
  if (rc) {
    tt_clvp_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_clvp_destroy(tt_clvp* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  if (e->guard_host) (void)hipHostFree(e->guard_host);
  e->arena.release();
  e->sb.destroy();
  delete e;
}

int tt_clvp_score(tt_clvp* e, const int* text, int T, const int* codes, int B, int n, float* scores, void* stream) {
  TT_REQUIRE(e && text && codes && scores, "tt_clvp_score: null argument");
  TT_REQUIRE(T >= 1 && n >= 8 && B >= 1 && (size_t)B * n <= (size_t)e->cfg.max_rows && T <= e->cfg.max_rows,
             "tt_clvp_score: T=%d B=%d n=%d exceed capacity %d rows (n must be >= 8)", T, B, n, e->cfg.max_rows);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  TT_TRY(clvp_tower_run(e, e->text, text, 1, T, e->text_latent, s));
  TT_TRY(clvp_tower_run(e, e->speech, codes, B, n, e->speech_latent, s));
  TT_TRY(clvp_score_launch(e->text_latent, 1, e->speech_latent, e->temperature, scores, B, e->cfg.latent_dim, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s));
  return e->sb.leave(us);
}

// Several utterances of one voice in ONE speech-tower pass (long-form reading: tortoise/read.py:66-71 scores its chunks one after the other,
// api.py:460-477 each time): utterance g has its own text (its tower pass is a few dozen rows) and candidates [g * N, (g + 1) * N) of `codes`.
// A candidate's score is the same bits as from tt_clvp_score on its utterance alone (row-local towers, batch-independent GEMM k order).
int tt_clvp_score_groups(tt_clvp* e, const int* texts, const int* T_host, int G, const int* codes, int N, int n, float* scores, void* stream) {
  TT_REQUIRE(e && texts && T_host && codes && scores, "tt_clvp_score_groups: null argument");
  TT_REQUIRE(G >= 1 && G <= 16 && N >= 1 && n >= 8 && (size_t)G * N * n <= (size_t)e->cfg.max_rows, "tt_clvp_score_groups: %d utterances x %d candidates x %d codes exceed capacity %d rows (<= 16 utterances, n >= 8)", G, N, n, e->cfg.max_rows);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int LD = e->cfg.latent_dim;
  size_t off = 0;
  for (int g = 0; g < G; ++g) {
    TT_REQUIRE(T_host[g] >= 1 && T_host[g] <= e->cfg.max_rows, "tt_clvp_score_groups: text %d has %d tokens", g, T_host[g]);
    TT_TRY(clvp_tower_run(e, e->text, texts + off, 1, T_host[g], e->text_latent + (size_t)g * LD, s));
    off += T_host[g];
  }
  TT_TRY(clvp_tower_run(e, e->speech, codes, G * N, n, e->speech_latent, s));
  for (int g = 0; g < G; ++g)
    TT_TRY(clvp_score_launch(e->text_latent + (size_t)g * LD, 1, e->speech_latent + (size_t)g * N * LD, e->temperature, scores + (size_t)g * N, N, LD, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s));
  return e->sb.leave(us);
}

// Operand-overflow guard of this stage (see tt_ar_guard), as of the last finished tt_clvp_score (after the caller synchronised).
int tt_clvp_guard(tt_clvp* e, int reset) {
  if (!e) { set_error("tt_clvp_guard: null handle"); return -1; }
  const int n = e->guard_host[0];
  if (n > 0) set_error("CLVP stage: %d kernel(s) met non-finite values (operand overflow in %s)", n, e->cfg.dtype == DT_F16 ? "fp16: use bf16 operands for this stage" : "bf16");
  if (reset && n > 0) {  // (a clean counter needs no device work: this sits at the end of every utterance)
    if (hipMemsetAsync(e->guard, 0, 4 * sizeof(int), e->sb.own) != hipSuccess || hipStreamSynchronize(e->sb.own) != hipSuccess) { set_error("tt_clvp_guard: reset failed"); return -2; }
    e->guard_host[0] = 0;
  }
  return n;
}

}  // extern "C"
