// Stage 1: UnifiedVoice's GPT-2 trunk (30 x 1024 x 16 heads) as an MI355X-native engine.
//   * prefill:   every candidate of an utterance shares the same [cond | text | start] prefix, so it is
//                evaluated once (M = P+1 rows) and its K/V are shared by all sequences
//                (the reference recomputes it B times: autoregressive.py:134-144 + repeat_interleave).
//   * decode:    KV-cached step for B sequences, seven launches per layer: LayerNorm (folds the split-K slabs + bias of the MLP projection
//                before it into the residual stream), QKV GEMM (K / V appended by its epilogue), decode attention, attention projection
//                (split-K slabs), LayerNorm (folds them), c_fc GEMM + gelu, MLP projection (slabs).  Sampling on device; the whole step
//                is replayed from one hipGraph.  Batches of <= 64 sequences (the per-rank share of a candidate-sharded job) run their
//                GEMMs on 16-column tiles so that the weight stream is spread over 192 - 512 workgroups (gemm_impl.h Tile).
//                (A five-launch form - LayerNorm folded into the GEMMs algebraically, split-K folded in-launch behind arrival tickets -
//                was built and measured 0.7 - 10 % slower at every batch size from 16 to 256: profiles/r05_ab_ar_five_launch_step.txt,
//                profiles/r06_ab_small_batch_decode.txt; it is not in the library.)
//   * latents:   teacher-forced full pass for the CLVP winners (autoregressive.py:454-506).
#include "runtime.h"
#include <unistd.h>
#include <chrono>
#include "../../include/tortoise_mi355x.h"

using namespace tt;

struct ArParams {
  unsigned long long keys[16];
  int row_offset;
  int pad[3];
};

struct tt_ar {
  tt_ar_config cfg;
  tt_ar_weights w;
  std::vector<tt_gpt_layer> L;
  int D, H, V;
  int es = 2;               // bytes per operand element (2: bf16 / fp16, 4: the fp32 verification mode)
  int Vp = 0;               // vocabulary padded to a multiple of 4: row stride of the logits and rows of the padded head copy
  void* w_head_p = nullptr; // [Vp][D] T  lm_head weight with zero rows appended (8194 -> 8196: every epilogue access of the head GEMM
  float* b_head_p = nullptr;//            is a whole aligned quad - the run-time-ragged generic kernel cost 24 us per step instead of ~12)
  Arena arena;
  StreamBridge sb;
  // shared prefix cache [layers][H][P1][64] (row-major) and per-sequence cache
  void* kp = nullptr; void* vp = nullptr;
  void* kc = nullptr; void* vc = nullptr;
  size_t prefix_layer_elems = 0, gen_layer_elems = 0;
  int tmax = 0;
  // activations
  float* x = nullptr;      // [rows][D] residual stream
  void* h = nullptr;       // [rows][D]  T
  void* ff = nullptr;      // [rows][4D] T
  void* attn = nullptr;    // [rows][D]  T
  void* q = nullptr;       // full pass: [BH][n][64]; decode: [B][D]
  void* kfull = nullptr;   // full pass scratch keys
  void* vt = nullptr;      // full pass V^T [BH][64][n_pad]
  float* slabs = nullptr;  // split-K partials [MAX_SPLIT][max_batch][D]
  float* logits = nullptr; // [max_batch][V]
  float* typ_logits = nullptr;  // [max_batch][Vp]: the rows the sampler reads under typical sampling (tt_sampling.typical_mass)
  int* state = nullptr; unsigned* seen = nullptr; int* unfinished = nullptr; int* unfinished_count = nullptr;
  int* next_tok = nullptr;
  int* guard = nullptr;    // [4] device counters: [0] rows with a non-finite value seen by the row norms / the sampler (tt_ar_guard)
  int* guard_host = nullptr;  // pinned copy, refreshed at the end of every generation / latent pass
  int max_rows = 0;
  int P1 = 0;      // current prefix length (incl. start token); with several groups the longest one
  int G = 1;       // utterances (groups) of the current batch, each with its own prefix: kp / vp are [group][layer][H][max_prefix][64]
  int P1g[16] = {0};
  unsigned prefilled = 0;  // bit g: group g's prefix has been evaluated for the current batch
  int B = 0;       // current batch
  int logits_rows = 0;
  bool logits_from_prefill = false;
  int host_slot = -1;  // host mirror of state[1]; only feeds the profiler's byte estimates
  // streaming (api_fast.py:389-414 consumes the (token, latent) pairs of the sampling loop): lm_head's input norm also files its f32
  // row - final_norm(ln_f(hidden)), the reference's per-step latent - under [latent index][sequence]; small batches only
  float* lat = nullptr;
  int lat_batch = 0;
  int gen_done = 0;    // tokens sampled by the running generation (tt_ar_generate / tt_ar_generate_chunk)
  bool gen_finished = false;
  // The captured decode step is kept between calls: everything a call can change is either device data (token / slot counters, the
  // Philox keys below, the prefix caches) or part of step_key - the bytes of the sampler's argument block plus the batch / group /
  // prefix-length values the launchers bake into the graph.  A call with the same key replays step_exec; any other key re-captures.
  ArParams* par_dev = nullptr;    // Philox key per utterance group (group 0 alone without groups) + row_offset of the call
  ArParams* par_host = nullptr;   // pinned staging of the same
  hipGraph_t step_graph = nullptr;
  hipGraphExec_t step_exec = nullptr;
  std::vector<unsigned char> step_key;
  // The sampler writes into a code buffer the HANDLE owns ([max_batch][tmax], stop-filled at the start of a generation) and the
  // finished columns are copied to the caller's buffer at the end of a call: the caller's pointer is not part of the kept graph.
  int* codes_own = nullptr;
  // Progress words in pinned host memory, written by the step's last kernel (system-scope stores): [0] tokens sampled so far,
  // [1] index of the first token after which every row had stopped (-1: none yet).  The host launches steps at most `lookahead`
  // ahead of [0] and stops launching when [1] turns >= 0: no queue drain inside the loop (round 3: a D2H copy + stream
  // synchronisation every 8 steps), at most `lookahead` surplus steps after the last row stopped.
  int* progress_host = nullptr;
  int* progress_dev = nullptr;
  int lookahead = 6;
  // Handles that decode at most 4 sequences (the streaming engine of api_fast.py: max_batch = 1): the decode step's GEMMs run GEMV-shaped
  // (gemv.hip) - a property of the HANDLE, not of a call's batch, so a handle's kernels never change between calls
  int gemv = 0;  // 0 | 1 GEMV launches | 2 GEMV launches that also do the layer norm in front of them (five launches per layer)
  int captures = 0;   // decode-step captures so far (tt_ar_stat: tests assert the kept graph is reused)
  int drains = 0;     // host-side queue drains the launch loop fell back to (0 when the progress words arrive)
  bool typical = false;  // the last generation ran the typical-sampling mask ahead of the sampler (one more launch per step)
};

namespace tt { int g_ar_gemv = 2; }  // ttx_kernel_variant(TTX_AR_GEMV), read at tt_ar_create: handles of <= 4 sequences run 0 = the MFMA decode GEMMs | 1 = GEMV launches | 2 = GEMVs with the layer norms inside

static void ar_drop_step_graph(tt_ar* e) {
  if (e->step_exec) (void)hipGraphExecDestroy(e->step_exec);
  if (e->step_graph) (void)hipGraphDestroy(e->step_graph);
  e->step_exec = nullptr;
  e->step_graph = nullptr;
  e->step_key.clear();
}

static const int MAX_SPLIT = 8;

static inline GemmArgs ar_gemm(const tt_ar* e, const void* A, int lda, const void* W, int ldw, int M, int N, int K) {
  return gemm_args(A, lda, W, ldw, M, N, K);
}

// x [M][D] rows (+ pending bias / split-K slabs of the GEMM before) -> LayerNorm -> h_out [M][D] T.  slabs: [nslab][slab_rows][D]
static int ar_rownorm_rows(tt_ar* e, float* x, void* h_out, int M, const float* g1, const float* b1, const float* add_bias, const float* slabs,
                           int nslab, int slab_rows, hipStream_t s) {
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = e->D; a.M = M; a.D = e->D;
  a.add_bias = add_bias;
  a.add_slabs = nslab ? slabs : nullptr;
  a.nslab = nslab; a.slab_stride = (size_t)slab_rows * e->D; a.ldslab = e->D;
  a.write_x = (add_bias || nslab) ? 1 : 0;
  a.mode = NORM_LAYER;
  a.g1 = g1; a.b1 = b1; a.eps1 = 1e-5f;
  a.out_t = h_out; a.ldot = e->D;
  a.row_blocks = 1;
  a.guard = e->guard;
  return rownorm_launch(e->cfg.dtype, a, s);
}
static int ar_rownorm(tt_ar* e, float* x, int M, const float* g1, const float* b1, const float* g2, const float* b2,
                      const float* add_bias, int nslab, int slab_rows, hipStream_t s) {
  (void)g2; (void)b2;
  return ar_rownorm_rows(e, x, e->h, M, g1, b1, add_bias, e->slabs, nslab, slab_rows, s);
}

// GPT2Model.forward over full sequences (causal), in place on e->x [B*n][D].
static int gpt_trunk_full(tt_ar* e, int B, int n, bool to_prefix, hipStream_t s, int group = 0) {
  const int D = e->D, H = e->H, M = B * n, dt = e->cfg.dtype;
  const int n_pad = round_up(n, 32);
  for (int l = 0; l < e->cfg.layers; ++l) {
    const tt_gpt_layer& w = e->L[l];
    TT_TRY(ar_rownorm(e, e->x, M, w.ln1_g, w.ln1_b, nullptr, nullptr, nullptr, 0, 0, s));
    GemmArgs g = ar_gemm(e, e->h, D, w.w_qkv, D, M, 3 * D, D);
    g.bias = w.b_qkv; g.seq_len = n; g.dmodel = D; g.heads = H;
    g.q = e->q;
    const size_t pl = ((size_t)group * e->cfg.layers + l) * e->prefix_layer_elems;
    g.k = to_prefix ? offset_t(e->kp, pl, e->es) : e->kfull;
    g.v = to_prefix ? offset_t(e->vp, pl, e->es) : nullptr;
    g.vt = e->vt; g.seq_pad = n_pad; g.q_scale = 0.125f;
    TT_TRY(gemm_launch(dt, EPI_QKV_HEADS, g, s));
    FlashArgs f;
    memset(&f, 0, sizeof(f));
    f.q = e->q; f.k = g.k; f.vt = e->vt; f.out = e->attn; f.ldo = D;
    f.BH = B * H; f.heads = H; f.n = n; f.n_pad = n_pad; f.causal = 1;
    TT_TRY(flash_attention_launch(dt, f, s));
    g = ar_gemm(e, e->attn, D, w.w_proj, D, M, D, D);
    g.bias = w.b_proj; g.res = e->x; g.ldres = D; g.out_f32 = e->x; g.ldo32 = D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    TT_TRY(ar_rownorm(e, e->x, M, w.ln2_g, w.ln2_b, nullptr, nullptr, nullptr, 0, 0, s));
    g = ar_gemm(e, e->h, D, w.w_fc, D, M, 4 * D, D);
    g.bias = w.b_fc; g.act = ACT_GELU_TANH; g.out_t = e->ff; g.ldot = 4 * D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    g = ar_gemm(e, e->ff, 4 * D, w.w_proj2, 4 * D, M, D, 4 * D);
    g.bias = w.b_proj2; g.res = e->x; g.ldres = D; g.out_f32 = e->x; g.ldo32 = D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  }
  return 0;
}

// Split-K factor of the decode projections.  Deliberately independent of the batch size: the k-order
// of every output element is then the same for any B, so a candidate's logits (and therefore its
// sampled codes) do not depend on how the candidates are sharded across GPUs.
static int pick_split(int B, int N, int K) {
  (void)B;
  // decode GEMMs run 64x64 tiles with 256-deep k-stages: aim for ~256 blocks at the full candidate batch
  // (4 row tiles) and at least one whole stage per split.  (In-situ A/B, round 3: 8 slabs for proj2 +4.8 %, 2 slabs for proj +0.9 %.)
  const int tiles = 4 * cdiv(N, 64);
  int sk = 256 / (tiles > 0 ? tiles : 1);
  const int nk = K / 256;
  if (sk > MAX_SPLIT) sk = MAX_SPLIT;
  if (sk > nk) sk = nk;
  if (sk < 1) sk = 1;
  return sk;
}

// lm_head = Sequential(final_norm, mel_head) applied to ln_f(x) (autoregressive.py:42, 174)
// lat_index: >= 0 files the normalised row(s) as that latent (prefill: 0); -1: under the device-side step counter (decode step
// feeding token i - 1 produces latent i); -2: no capture
static int ar_head_norm(tt_ar* e, float* x, void* h_out, int M, const float* add_bias, const float* slabs, int nslab, int slab_rows,
                        hipStream_t s, int lat_index = -2) {
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = e->D; a.M = M; a.D = e->D;
  a.add_bias = add_bias;
  a.add_slabs = nslab ? slabs : nullptr;
  a.nslab = nslab; a.slab_stride = (size_t)slab_rows * e->D; a.ldslab = e->D;
  a.write_x = (add_bias || nslab) ? 1 : 0;
  a.mode = NORM_LAYER;
  a.g1 = e->w.lnf_g; a.b1 = e->w.lnf_b; a.eps1 = 1e-5f;
  a.g2 = e->w.final_norm_g; a.b2 = e->w.final_norm_b; a.eps2 = 1e-5f;
  a.out_t = h_out; a.ldot = e->D;
  a.row_blocks = 1;
  a.guard = e->guard;
  if (e->lat && lat_index != -2 && M <= e->lat_batch) {
    a.out_f32 = e->lat; a.ldo32 = e->D;
    a.f32_slot_stride = (size_t)e->lat_batch * e->D;
    if (lat_index >= 0) a.out_f32 += (size_t)lat_index * a.f32_slot_stride;
    else { a.f32_slot = e->state + 1; a.f32_slot_base = 1; }
  }
  return rownorm_launch(e->cfg.dtype, a, s);
}
static int ar_head_gemm(tt_ar* e, int M, hipStream_t s, int logits_row0 = 0) {
  if (e->gemv && M <= 4) {
    GemvArgs v;
    memset(&v, 0, sizeof(v));
    v.A = e->h; v.lda = e->D; v.W = e->w_head_p; v.ldw = e->D; v.M = M; v.N = e->Vp; v.K = e->D; v.bias = e->b_head_p; v.epi = GEMV_F32;
    v.out_f32 = e->logits + (size_t)logits_row0 * e->Vp; v.ldo32 = e->Vp;
    TT_TRY(gemv_launch(e->cfg.dtype, v, s));
    e->logits_rows = logits_row0 + M;
    return 0;
  }
  GemmArgs g = ar_gemm(e, e->h, e->D, e->w_head_p, e->D, M, e->Vp, e->D);
  g.bias = e->b_head_p; g.out_f32 = e->logits + (size_t)logits_row0 * e->Vp; g.ldo32 = e->Vp;
  TT_TRY(gemm_launch(e->cfg.dtype, EPI_STD, g, s));
  e->logits_rows = logits_row0 + M;
  return 0;
}
static int ar_head(tt_ar* e, float* x, int M, const float* add_bias, int nslab, hipStream_t s, int lat_index = -2, int logits_row0 = 0) {
  TT_TRY(ar_head_norm(e, x, e->h, M, add_bias, e->slabs, nslab, e->B, s, lat_index));
  return ar_head_gemm(e, M, s, logits_row0);
}

// The 30 layers of one KV-cached decode step for the e->B sequences + the input norm of lm_head, all on stream s.
static int decode_layers_enqueue(tt_ar* e, hipStream_t s) {
  const int D = e->D, H = e->H, dt = e->cfg.dtype, nb = e->B;
  float* x = e->x;
  float* slabs = e->slabs;
  const float* pend_bias = nullptr;
  int pend_slabs = 0;
  for (int l = 0; l < e->cfg.layers; ++l) {
    const tt_gpt_layer& w = e->L[l];
    if (e->gemv < 2) TT_TRY(ar_rownorm_rows(e, x, e->h, nb, w.ln1_g, w.ln1_b, pend_bias, slabs, pend_slabs, nb, s));
    if (e->gemv) {  // <= 4 sequences: GEMV-shaped launches, no split-K - the projections update x in place, the norms fold nothing
      GemvArgs v;
      memset(&v, 0, sizeof(v));
      v.A = e->h; v.lda = D; v.W = w.w_qkv; v.ldw = D; v.M = nb; v.N = 3 * D; v.K = D; v.bias = w.b_qkv; v.epi = GEMV_QKV;
      v.step = e->state + 1; v.qbuf = e->q; v.kc = offset_t(e->kc, (size_t)l * e->gen_layer_elems, e->es); v.vc = offset_t(e->vc, (size_t)l * e->gen_layer_elems, e->es);
      v.heads = H; v.tmax = e->tmax; v.dmodel = D; v.q_scale = 0.125f;
      if (e->gemv == 2) { v.ln_x = x; v.ldx = D; v.ln_g = w.ln1_g; v.ln_b = w.ln1_b; v.ln_eps = 1e-5f; v.guard = e->guard; }
      TT_TRY(gemv_launch(dt, v, s));
      DecodeAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = e->q;
      a.kp = offset_t(e->kp, (size_t)l * e->prefix_layer_elems, e->es);
      a.vp = offset_t(e->vp, (size_t)l * e->prefix_layer_elems, e->es);
      a.P1 = e->P1; a.kc = v.kc; a.vc = v.vc; a.tmax = e->tmax; a.step = e->state + 1;
      a.out = e->attn; a.B = nb; a.heads = H; a.host_tgen = e->host_slot + 1;
      TT_TRY(decode_attention_launch(dt, a, s));
      memset(&v, 0, sizeof(v));
      v.A = e->attn; v.lda = D; v.W = w.w_proj; v.ldw = D; v.M = nb; v.N = D; v.K = D; v.bias = w.b_proj; v.epi = GEMV_RES; v.out_f32 = x; v.ldo32 = D;
      TT_TRY(gemv_launch(dt, v, s));
      if (e->gemv < 2) TT_TRY(ar_rownorm_rows(e, x, e->h, nb, w.ln2_g, w.ln2_b, nullptr, slabs, 0, nb, s));
      memset(&v, 0, sizeof(v));
      if (e->gemv == 2) { v.ln_x = x; v.ldx = D; v.ln_g = w.ln2_g; v.ln_b = w.ln2_b; v.ln_eps = 1e-5f; v.guard = e->guard; }
      v.A = e->h; v.lda = D; v.W = w.w_fc; v.ldw = D; v.M = nb; v.N = 4 * D; v.K = D; v.bias = w.b_fc; v.epi = GEMV_GELU_T; v.out_t = e->ff; v.ldot = 4 * D;
      TT_TRY(gemv_launch(dt, v, s));
      memset(&v, 0, sizeof(v));
      v.A = e->ff; v.lda = 4 * D; v.W = w.w_proj2; v.ldw = 4 * D; v.M = nb; v.N = D; v.K = 4 * D; v.bias = w.b_proj2; v.epi = GEMV_RES; v.out_f32 = x; v.ldo32 = D;
      TT_TRY(gemv_launch(dt, v, s));
      pend_bias = nullptr;
      pend_slabs = 0;
      continue;
    }
    GemmArgs g = ar_gemm(e, e->h, D, w.w_qkv, D, nb, 3 * D, D);
    g.bias = w.b_qkv;
    g.dmodel = D; g.heads = H; g.q_scale = 0.125f;
    g.step = e->state + 1; g.qbuf = e->q;
    g.kc = offset_t(e->kc, (size_t)l * e->gen_layer_elems, e->es);
    g.vc = offset_t(e->vc, (size_t)l * e->gen_layer_elems, e->es);
    g.tmax = e->tmax;
    TT_TRY(gemm_launch(dt, EPI_QKV_DECODE, g, s));
    DecodeAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = e->q;
    a.kp = offset_t(e->kp, (size_t)l * e->prefix_layer_elems, e->es);
    a.vp = offset_t(e->vp, (size_t)l * e->prefix_layer_elems, e->es);
    if (e->G > 1) {
      a.ngroups = e->G; a.group_size = nb / e->G;
      a.prefix_group_stride = (size_t)e->cfg.layers * e->prefix_layer_elems;
      for (int gi = 0; gi < e->G; ++gi) a.p1_tab[gi] = e->P1g[gi];
    }
    a.P1 = e->P1; a.kc = g.kc; a.vc = g.vc; a.tmax = e->tmax; a.step = e->state + 1;
    a.out = e->attn; a.B = nb; a.heads = H; a.host_tgen = e->host_slot + 1;
    TT_TRY(decode_attention_launch(dt, a, s));
    // >= 1024 sequences (several utterances per batch): one block per output tile fills the chip, so the split-K partial
    // sums are folded inside the launch in slab order (gemm.h serial_k: the same bits as slabs + row norm, without the slab traffic)
    const bool serial = nb >= 1024;
    int sk = pick_split(nb, D, D);
    g = ar_gemm(e, e->attn, D, w.w_proj, D, nb, D, D);
    if (serial && sk > 1) {
      g.serial_k = sk; g.bias = w.b_proj; g.res = x; g.ldres = D; g.out_f32 = x; g.ldo32 = D;
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
      TT_TRY(ar_rownorm_rows(e, x, e->h, nb, w.ln2_g, w.ln2_b, nullptr, slabs, 0, nb, s));
    } else {
      g.splitk = sk; g.out_f32 = slabs; g.ldo32 = D;
      if (sk == 1) { g.bias = nullptr; }
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
      TT_TRY(ar_rownorm_rows(e, x, e->h, nb, w.ln2_g, w.ln2_b, w.b_proj, slabs, sk, nb, s));
    }
    g = ar_gemm(e, e->h, D, w.w_fc, D, nb, 4 * D, D);
    g.bias = w.b_fc; g.act = ACT_GELU_TANH; g.out_t = e->ff; g.ldot = 4 * D;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    sk = pick_split(nb, D, 4 * D);
    g = ar_gemm(e, e->ff, 4 * D, w.w_proj2, 4 * D, nb, D, 4 * D);
    if (serial && sk > 1) {
      g.serial_k = sk; g.bias = w.b_proj2; g.res = x; g.ldres = D; g.out_f32 = x; g.ldo32 = D;
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
      pend_bias = nullptr;
      pend_slabs = 0;
    } else {
      g.splitk = sk; g.out_f32 = slabs; g.ldo32 = D;
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
      pend_bias = w.b_proj2;
      pend_slabs = sk;
    }
  }
  return ar_head_norm(e, x, e->h, nb, pend_bias, slabs, pend_slabs, nb, s, -1);
}

// One KV-cached decode step for e->B sequences up to the logits; the fed tokens are in e->next_tok (or, `embedded`, the sampler
// already wrote this step's input rows into e->x: tt_ar_generate).
static int decode_step_enqueue(tt_ar* e, hipStream_t s, bool embedded = false) {
  const int B = e->B;
  if (!embedded) TT_TRY(ar_embed_launch(e->next_tok, e->state, e->w.mel_emb, e->w.mel_pos, e->x, B, e->D, e->cfg.mel_pos_offset, s));
  TT_TRY(decode_layers_enqueue(e, s));
  return ar_head_gemm(e, B, s);
}

// Capture fn() on stream s into a new graph + executable.
template <typename F>
static int ar_capture(hipStream_t s, F&& fn, hipGraph_t* graph_out, hipGraphExec_t* exec_out) {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  TT_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = fn();
  hipError_t ce = hipStreamEndCapture(s, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (ce == hipSuccess) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ce != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    set_error("tt_ar_generate: graph capture / instantiate failed: %s", hipGetErrorString(ce));
    return -2;
  }
  *graph_out = graph;
  *exec_out = exec;
  return 0;
}

extern "C" {

int tt_ar_create(const tt_ar_config* cfg, const tt_ar_weights* w, tt_ar** out) {
  TT_REQUIRE(cfg && w && out, "tt_ar_create: null argument");
  TT_REQUIRE(cfg->heads * 64 == cfg->model_dim, "tt_ar_create: head_dim must be 64 (model_dim=%d heads=%d)", cfg->model_dim, cfg->heads);
  TT_REQUIRE(cfg->model_dim % 64 == 0 && cfg->model_dim <= 4096, "tt_ar_create: unsupported model_dim %d", cfg->model_dim);
  TT_REQUIRE(cfg->max_batch > 0 && cfg->max_prefix > 1 && cfg->max_new_tokens > 0, "tt_ar_create: bad capacity");
  TT_REQUIRE(cfg->vocab <= 10240, "tt_ar_create: vocab %d exceeds the sampler's 10240 limit", cfg->vocab);
  TT_REQUIRE(cfg->mel_pos_offset == 1 || cfg->mel_pos_offset == 2, "tt_ar_create: mel_pos_offset must be 2 (kv_cache=True rule) or 1 (kv_cache=False rule), got %d", cfg->mel_pos_offset);
  TT_REQUIRE(cfg->max_groups >= 0 && cfg->max_groups <= 16, "tt_ar_create: max_groups %d outside 0 .. 16", cfg->max_groups);
  TT_REQUIRE(cfg->dtype == DT_BF16 || cfg->dtype == DT_F16 || cfg->dtype == DT_F32, "tt_ar_create: unknown dtype %d", cfg->dtype);
  TT_REQUIRE(cfg->dtype != DT_F32 || cfg->max_groups <= 1, "tt_ar_create: the fp32 verification mode decodes one utterance per batch");
  tt_ar* e = new tt_ar();
  e->cfg = *cfg;
  if (e->cfg.max_groups < 1) e->cfg.max_groups = 1;
  e->w = *w;
  e->L.assign(w->layers_host, w->layers_host + cfg->layers);
  e->D = cfg->model_dim; e->H = cfg->heads; e->V = cfg->vocab;
  e->es = dtype_bytes(cfg->dtype);
  const size_t es = e->es;
  // KV slots per sequence, rounded up to 8: the K cache is chunk-major with a chunk stride of 16 tmax bytes, and strides within ~100 bytes of a
  // multiple of 4 KB (tmax = 254, 506, 508, 510, 516 ...) cost the decode attention up to 30 % (scripts/attn_stride.py, profiles/r06_attn_phase_stamps.txt);
  // a multiple of 8 slots is a multiple of 128 bytes and cannot fall inside that band
#ifdef TT_KV_NO_ROUND  // (A/B builds)
  e->tmax = cfg->max_new_tokens;
#else
  e->tmax = round_up(cfg->max_new_tokens, 8);
#endif
  const int D = e->D, H = e->H;
  int rc = e->sb.init();
  e->max_rows = std::max(std::max(cfg->max_prefix, cfg->max_full_rows), cfg->max_batch);
  const size_t rows = (size_t)e->max_rows + 64;
  e->prefix_layer_elems = (size_t)H * cfg->max_prefix * 64;
  e->gen_layer_elems = (size_t)cfg->max_batch * H * e->tmax * 64;
  const int npad_max = round_up(e->max_rows, 32) + 32;
  if (!rc) rc = e->arena.alloc(&e->kp, (e->prefix_layer_elems * cfg->layers * e->cfg.max_groups + 4096) * es);
  if (!rc) rc = e->arena.alloc(&e->vp, (e->prefix_layer_elems * cfg->layers * e->cfg.max_groups + 4096) * es);
  if (!rc) rc = e->arena.alloc(&e->kc, (e->gen_layer_elems * cfg->layers + 4096) * es);
  if (!rc) rc = e->arena.alloc(&e->vc, (e->gen_layer_elems * cfg->layers + 4096) * es);
  if (!rc) rc = e->arena.alloc_t(&e->x, rows * D);
  if (!rc) rc = e->arena.alloc(&e->h, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->ff, rows * 4 * D * es);
  if (!rc) rc = e->arena.alloc(&e->attn, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->q, rows * D * es);
  if (!rc) rc = e->arena.alloc(&e->kfull, rows * D * es);
  // V^T scratch: per (batch, head) 64 rows of n_pad keys; total keys <= max_rows (+ padding per sequence)
  if (!rc) rc = e->arena.alloc(&e->vt, ((size_t)H * 64 * ((size_t)npad_max + 32 * (size_t)cfg->max_batch)) * es);
  if (!rc) rc = e->arena.alloc_t(&e->slabs, (size_t)MAX_SPLIT * cfg->max_batch * D);
  e->Vp = round_up(e->V, 4);
  if (!rc) rc = e->arena.alloc_t(&e->logits, (size_t)cfg->max_batch * e->Vp);
  if (!rc) rc = e->arena.alloc_t(&e->typ_logits, (size_t)cfg->max_batch * e->Vp);
  if (!rc) rc = e->arena.alloc(&e->w_head_p, (size_t)e->Vp * D * es);   // (arena memory is zeroed: the padding rows / bias entries are 0)
  if (!rc) rc = e->arena.alloc_t(&e->b_head_p, e->Vp);
  if (!rc && hipMemcpy(e->w_head_p, w->w_mel_head, (size_t)e->V * D * es, hipMemcpyDeviceToDevice) != hipSuccess) { set_error("tt_ar_create: copying the head weight failed"); rc = -2; }
  if (!rc && hipMemcpy(e->b_head_p, w->b_mel_head, (size_t)e->V * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) { set_error("tt_ar_create: copying the head bias failed"); rc = -2; }
  if (!rc) rc = e->arena.alloc_t(&e->state, 4);
  if (!rc) rc = e->arena.alloc_t(&e->seen, (size_t)cfg->max_batch * ((e->V + 31) / 32));
  if (!rc) rc = e->arena.alloc_t(&e->unfinished, cfg->max_batch);
  if (!rc) rc = e->arena.alloc_t(&e->unfinished_count, e->tmax + 8);
  if (!rc) rc = e->arena.alloc_t(&e->next_tok, cfg->max_batch);
  if (!rc && cfg->max_batch <= 8) {  // the streaming path decodes one sequence (api_fast.py): [tmax + 1][max_batch][D] f32
    e->lat_batch = cfg->max_batch;
    rc = e->arena.alloc_t(&e->lat, (size_t)(e->tmax + 1) * e->lat_batch * D);
  }
  if (!rc) rc = e->arena.alloc_t(&e->par_dev, 1);
  if (!rc) rc = e->arena.alloc_t(&e->codes_own, (size_t)cfg->max_batch * e->tmax);
  if (!rc) rc = e->arena.alloc_t(&e->guard, 4);
  if (!rc && (hipHostMalloc((void**)&e->par_host, sizeof(ArParams)) != hipSuccess ||
              hipHostMalloc((void**)&e->progress_host, 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
              hipHostMalloc((void**)&e->guard_host, 4 * sizeof(int)) != hipSuccess ||
              hipHostGetDevicePointer((void**)&e->progress_dev, e->progress_host, 0) != hipSuccess)) {
    set_error("tt_ar_create: hipHostMalloc failed");
    rc = -2;
  }
  if (!rc) {
    e->guard_host[0] = 0;
    e->progress_host[0] = 0; e->progress_host[1] = -1; e->progress_host[2] = e->progress_host[3] = 0;
  }
  // GEMV-shaped decode step: <= 4 sequences per handle, 16-bit operands, the trunk's K in {1024, 2048, 4096} (gemv.hip)
  e->gemv = (cfg->max_batch <= 4 && cfg->dtype != DT_F32 && cfg->max_groups <= 1 && D == 1024) ? tt::g_ar_gemv : 0;
  if (rc) {
    tt_ar_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_ar_destroy(tt_ar* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  ar_drop_step_graph(e);
  if (e->par_host) (void)hipHostFree(e->par_host);
  if (e->progress_host) (void)hipHostFree(e->progress_host);
  if (e->guard_host) (void)hipHostFree(e->guard_host);
  e->arena.release();
  e->sb.destroy();
  delete e;
}

int tt_ar_prefill_group(tt_ar* e, int group, int n_groups, const float* prefix_emb, int P, void* stream) {
  TT_REQUIRE(e && prefix_emb, "tt_ar_prefill: null argument");
  TT_REQUIRE(n_groups >= 1 && n_groups <= e->cfg.max_groups && group >= 0 && group < n_groups, "tt_ar_prefill_group: group %d of %d (capacity %d groups)", group, n_groups, e->cfg.max_groups);
  TT_REQUIRE(P >= 1 && P + 1 <= e->cfg.max_prefix, "tt_ar_prefill: prefix of %d rows exceeds capacity %d", P + 1, e->cfg.max_prefix);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int D = e->D;
  if (n_groups != e->G || group == 0) {  // a new batch starts with its group 0
    e->G = n_groups;
    e->prefilled = 0;
  }
  const int P1 = P + 1;
  e->P1g[group] = P1;
  e->prefilled |= 1u << group;
  e->P1 = 0;
  for (int gi = 0; gi < n_groups; ++gi)
    if ((e->prefilled >> gi) & 1u) e->P1 = std::max(e->P1, e->P1g[gi]);
  // per-layer stride of the prefix cache = its capacity (H * max_prefix * 64): group g, layer l at (g * layers + l) strides
  TT_CHECK_HIP(hipMemcpyAsync(e->x, prefix_emb, (size_t)P * D * sizeof(float), hipMemcpyDeviceToDevice, s));
  // start-token row: mel_embedding[start] + mel_pos_embedding[0]  (autoregressive.py:137-141)
  {
    RowNormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = e->x + (size_t)P * D; a.ldx = D; a.M = 1; a.D = D;
    a.x_in = e->w.mel_emb + (size_t)e->cfg.start_mel_token * D; a.ldxin = D;
    a.add_bias = e->w.mel_pos;  // row 0
    a.write_x = 1; a.mode = NORM_NONE;
    TT_TRY(rownorm_launch(e->cfg.dtype, a, s));
  }
  TT_TRY(gpt_trunk_full(e, 1, P1, true, s, group));
  const int B_saved = e->B;
  e->B = 1;
  int rc = ar_head(e, e->x + (size_t)P * D, 1, nullptr, 0, s, group == 0 ? 0 : -2, group);  // logits row `group`
  e->B = B_saved;
  TT_TRY(rc);
  e->logits_from_prefill = e->prefilled == (n_groups >= 32 ? ~0u : (1u << n_groups) - 1u);
  return e->sb.leave(us);
}

int tt_ar_prefill(tt_ar* e, const float* prefix_emb, int P, void* stream) { return tt_ar_prefill_group(e, 0, 1, prefix_emb, P, stream); }

int tt_ar_get_logits(tt_ar* e, float* dst, int rows, void* stream) {
  TT_REQUIRE(e && dst && rows >= 1 && rows <= e->logits_rows, "tt_ar_get_logits: %d rows requested, %d available", rows, e ? e->logits_rows : 0);
  hipStream_t us = (hipStream_t)stream;
  TT_TRY(e->sb.enter(us));
  TT_CHECK_HIP(hipMemcpy2DAsync(dst, (size_t)e->V * sizeof(float), e->logits, (size_t)e->Vp * sizeof(float), (size_t)e->V * sizeof(float), (size_t)rows,
                                hipMemcpyDeviceToDevice, e->sb.own));
  return e->sb.leave(us);
}

int tt_ar_begin(tt_ar* e, int B, void* stream) {
  TT_REQUIRE(e && B >= 1 && B <= e->cfg.max_batch, "tt_ar_begin: batch %d exceeds capacity", B);
  TT_REQUIRE(e->P1 > 0, "tt_ar_begin: call tt_ar_prefill first");
  hipStream_t us = (hipStream_t)stream;
  TT_TRY(e->sb.enter(us));
  e->B = B;
  e->host_slot = -1;
  TT_TRY(ar_begin_launch(e->state, e->seen, e->unfinished, e->unfinished_count, B, e->V, e->tmax + 8, e->cfg.start_mel_token, e->sb.own));
  return e->sb.leave(us);
}

int tt_ar_decode_step(tt_ar* e, const int* tokens, void* stream) {
  TT_REQUIRE(e && tokens && e->B > 0, "tt_ar_decode_step: call tt_ar_begin first");
  // the step about to run writes KV slot host_slot + 1 and reads mel position row host_slot + 1 + mel_pos_offset
  TT_REQUIRE(e->host_slot + 1 < e->tmax, "tt_ar_decode_step: all %d KV slots of this handle are used", e->tmax);
  TT_REQUIRE(e->host_slot + 1 + e->cfg.mel_pos_offset < e->cfg.mel_pos_len, "tt_ar_decode_step: step %d is beyond the mel position table (%d rows)", e->host_slot + 1, e->cfg.mel_pos_len);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  TT_CHECK_HIP(hipMemcpyAsync(e->next_tok, tokens, (size_t)e->B * sizeof(int), hipMemcpyDeviceToDevice, s));
  TT_TRY(ar_state_advance_launch(e->state, nullptr, nullptr, s));  // state[1] = slot of the token being fed
  e->host_slot += 1;
  e->logits_from_prefill = false;
  TT_TRY(decode_step_enqueue(e, s));
  return e->sb.leave(us);
}

// Sampling loop shared by tt_ar_generate (fresh = true: the whole utterance) and tt_ar_generate_chunk (streaming: the loop
// is resumed where the previous chunk stopped; every per-step quantity lives in device-side state, so resuming is just
// replaying the step graph again).  Tokens [e->gen_done, target) are produced; codes is the caller's [B][ldcodes] buffer.
static int ar_generate_run(tt_ar* e, int B, bool fresh, int target, int ldcodes, const tt_sampling* sp, int* codes, int* n_steps_host,
                           int* finished_host, hipStream_t s) {
  TT_REQUIRE(sp->typical_mass == 0.f || (sp->typical_mass > 0.f && sp->typical_mass < 1.f), "tt_ar_generate: typical_mass %g outside (0, 1) (0 = off)",
             (double)sp->typical_mass);
  SampleArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.B = B; sa.V = e->V; sa.seen = e->seen;
  sa.rep_penalty = sp->repetition_penalty; sa.temperature = sp->temperature; sa.top_p = sp->top_p; sa.top_k = sp->top_k;
  sa.exp_noise = sp->exp_noise;
  sa.state = e->state; sa.unfinished = e->unfinished; sa.stop_token = e->cfg.stop_mel_token;
  sa.codes = e->codes_own; sa.ldcodes = e->tmax; sa.next_tok = e->next_tok; sa.unfinished_count = e->unfinished_count;
  sa.embed_x = e->x; sa.tok_emb = e->w.mel_emb; sa.pos_emb = e->w.mel_pos; sa.D = e->D; sa.pos_offset = e->cfg.mel_pos_offset;
  sa.pos_len = e->cfg.mel_pos_len;
  sa.guard = e->guard;
  sa.typical_mass = sp->typical_mass; sa.typical_out = e->typ_logits;
  e->typical = sp->typical_mass != 0.f;
  // the Philox keys and the row offset go through device memory (sa.seed / sa.group_seeds / sa.row_offset stay zero): neither the
  // seed nor the candidate range of a call is part of the step graph
  sa.seed = 0;
  sa.row_offset = 0;
  sa.keys_dev = e->par_dev->keys;
  sa.row_offset_dev = &e->par_dev->row_offset;
  for (int gi = 0; gi < 16; ++gi) e->par_host->keys[gi] = sp->seed;
  e->par_host->row_offset = sp->row_offset;
  if (e->G > 1) {
    TT_REQUIRE(B % e->G == 0 && (B / e->G) % 4 == 0, "tt_ar_generate: %d sequences do not split into %d groups of a multiple of 4", B, e->G);
    sa.ngroups = e->G; sa.group_size = B / e->G;
    for (int gi = 0; gi < e->G; ++gi) e->par_host->keys[gi] = sp->group_seeds ? sp->group_seeds[gi] : sp->seed;
  }
  // (par_host and the progress words are free again: every earlier generation ended with a stream synchronisation)
  TT_CHECK_HIP(hipMemcpyAsync(e->par_dev, e->par_host, sizeof(ArParams), hipMemcpyHostToDevice, s));
  volatile int* prog = e->progress_host;
  if (fresh) {
    e->B = B;
    e->gen_done = 0;
    e->gen_finished = false;
    prog[0] = 0;
    prog[1] = -1;
    TT_TRY(ar_begin_launch(e->state, e->seen, e->unfinished, e->unfinished_count, B, e->V, e->tmax + 8, e->cfg.start_mel_token, s));
    TT_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)codes, e->cfg.stop_mel_token, (size_t)B * ldcodes, s));
    TT_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)e->codes_own, e->cfg.stop_mel_token, (size_t)B * e->tmax, s));
    // token 0: every row samples from the shared prefill logits
    sa.logits = e->logits; sa.ldl = 0; sa.ldg = e->Vp;
    TT_REQUIRE(e->logits_from_prefill, "tt_ar_generate: the logits buffer does not hold the prefill logits of all %d group(s); call tt_ar_prefill / tt_ar_prefill_group first", e->G);
    e->logits_from_prefill = false;
    TT_TRY(sample_launch(sa, s));
    TT_TRY(ar_state_advance_launch(e->state, e->unfinished_count, e->progress_dev, s));
    e->gen_done = 1;
  } else {
    prog[0] = e->gen_done;
  }
  sa.logits = e->logits; sa.ldl = e->Vp; sa.ldg = e->Vp;  // (ldg is unused with per-row logits; set so that fresh and resumed runs share one step key)
  if (!fresh) {
    // Resumed chunk (streaming): between two chunks the caller may have run tt_ar_latents / tt_ar_prefill, which use e->x as
    // their residual stream, so the input row the sampler fused into e->x for the next step is gone.  Rebuild it from the
    // device-side state (newest token, its index): mel_embedding[next_tok] + mel_pos_embedding[state[1] + offset].
    TT_TRY(ar_embed_launch(e->next_tok, e->state, e->w.mel_emb, e->w.mel_pos, e->x, B, e->D, e->cfg.mel_pos_offset, s));
  }

  const bool use_graph = graphs_enabled() && target - e->gen_done > 1;
  int rc = 0;
  auto tail_enqueue = [&](hipStream_t st) -> int {
    TT_TRY(sample_launch(sa, st));
    return ar_state_advance_launch(e->state, e->unfinished_count, e->progress_dev, st);
  };
  if (use_graph) {
    // what the captured step bakes in besides device pointers owned by the handle: the sampler's argument block (scalars, the optional
    // injected-noise pointer) and the batch geometry / range structure the launchers read from the handle
    std::vector<unsigned char> key(sizeof(sa) + 24 * sizeof(int));
    memcpy(key.data(), &sa, sizeof(sa));
    int geo[24] = {B, e->G, e->P1, g_prof_on ? 1 : 0};
    for (int gi = 0; gi < 16; ++gi) geo[4 + gi] = gi < e->G ? e->P1g[gi] : 0;
    memcpy(key.data() + sizeof(sa), geo, sizeof(geo));
    if (e->step_exec == nullptr || key != e->step_key) {
      ar_drop_step_graph(e);
      rc = ar_capture(s, [&]() -> int {
        TT_TRY(decode_step_enqueue(e, s, true));
        return tail_enqueue(s);
      }, &e->step_graph, &e->step_exec);
      if (rc) {
        ar_drop_step_graph(e);
        return rc;
      }
      e->step_key.swap(key);
      e->captures += 1;
    }
  }
  const int first_step = e->gen_done;
  int steps_done = e->gen_done;
  bool stop_seen = false;
  for (int step = first_step; step < target; ++step) {
    // stay at most `lookahead` steps ahead of the device; the words are written by the last kernel of every step
    const auto wait0 = std::chrono::steady_clock::now();
    while (step - prog[0] >= e->lookahead && prog[1] < 0) {
      // 200 ms of wall time without the expected progress (not a spin count: a slow first replay - code-object load, a profiler - must
      // not look like lost progress words): fall back to a queue drain, always correct, only slower - and take the progress from the
      // DEVICE state the words mirror, so a host that cannot see the mapped words still paces and ends the loop correctly
      if (std::chrono::steady_clock::now() - wait0 > std::chrono::milliseconds(200)) {
        int st3[3] = {0, 0, -1};
        hipError_t ce = hipStreamSynchronize(s);
        if (ce == hipSuccess) ce = hipMemcpy(st3, e->state, sizeof(st3), hipMemcpyDeviceToHost);
        if (ce != hipSuccess) { set_error("tt_ar_generate: %s", hipGetErrorString(ce)); rc = -2; break; }
        prog[0] = st3[0];
        if (st3[2] >= 0) prog[1] = st3[2];
        e->drains += 1;
        break;
      }
      usleep(50);
    }
    if (rc) break;
    if (prog[1] >= 0) { stop_seen = true; break; }
    hipError_t le = hipSuccess;
    if (use_graph) {
      le = hipGraphLaunch(e->step_exec, s);
    } else {
      e->host_slot = step - 1;
      rc = decode_step_enqueue(e, s, true);
      if (!rc) rc = tail_enqueue(s);
      if (rc) break;
    }
    if (le != hipSuccess) { set_error("tt_ar_generate: step launch: %s", hipGetErrorString(le)); rc = -2; break; }
    steps_done = step + 1;
  }
  (void)stop_seen;
  if (!rc) {
    // the finished columns go to the caller's buffer; one host-visible completion point per call
    hipError_t ce = hipMemcpy2DAsync(codes, (size_t)ldcodes * sizeof(int), e->codes_own, (size_t)e->tmax * sizeof(int),
                                     (size_t)steps_done * sizeof(int), (size_t)B, hipMemcpyDeviceToDevice, s);
    if (ce == hipSuccess) ce = hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s);
    if (ce == hipSuccess) ce = hipStreamSynchronize(s);
    if (ce != hipSuccess) { set_error("tt_ar_generate: %s", hipGetErrorString(ce)); rc = -2; }
  }
  if (rc) {
    (void)hipStreamSynchronize(s);
    ar_drop_step_graph(e);  // a failed replay leaves nothing to trust
  }
  TT_TRY(rc);
  const int first_zero = prog[1];
  const bool finished = first_zero >= 0;
  e->gen_done = finished ? (first_zero + 1 < steps_done ? first_zero + 1 : steps_done) : steps_done;
  e->gen_finished = finished;
  e->host_slot = e->gen_done - 1;
  if (n_steps_host) *n_steps_host = e->gen_done;
  if (finished_host) *finished_host = finished ? 1 : 0;
  return 0;
}

int tt_ar_generate(tt_ar* e, int B, int max_new, const tt_sampling* sp, int* codes, int* n_steps_host, void* stream) {
  TT_REQUIRE(e && sp && codes, "tt_ar_generate: null argument");
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_batch, "tt_ar_generate: batch %d exceeds capacity %d", B, e->cfg.max_batch);
  TT_REQUIRE(max_new >= 1 && max_new <= e->tmax, "tt_ar_generate: max_new %d exceeds capacity %d", max_new, e->tmax);
  TT_REQUIRE(max_new - 2 + e->cfg.mel_pos_offset < e->cfg.mel_pos_len, "tt_ar_generate: max_new %d exceeds the mel position table", max_new);
  TT_REQUIRE(e->P1 > 0, "tt_ar_generate: call tt_ar_prefill first");
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  TT_TRY(ar_generate_run(e, B, true, max_new, max_new, sp, codes, n_steps_host, nullptr, s));
  return e->sb.leave(us);
}

int tt_ar_generate_chunk(tt_ar* e, int B, int first, int n_more, int ldcodes, const tt_sampling* sp, int* codes, int* n_total_host,
                         int* finished_host, void* stream) {
  TT_REQUIRE(e && sp && codes && n_total_host && finished_host, "tt_ar_generate_chunk: null argument");
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_batch, "tt_ar_generate_chunk: batch %d exceeds capacity %d", B, e->cfg.max_batch);
  TT_REQUIRE(e->P1 > 0, "tt_ar_generate_chunk: call tt_ar_prefill first");
  TT_REQUIRE(first || (e->gen_done >= 1 && e->B == B), "tt_ar_generate_chunk: no generation of %d rows to resume", B);
  const int done = first ? 0 : e->gen_done;
  const int target = done + n_more;
  TT_REQUIRE(n_more >= 1 && target <= ldcodes && target <= e->tmax, "tt_ar_generate_chunk: %d + %d tokens exceed capacity (%d code columns, %d KV slots)", done, n_more, ldcodes, e->tmax);
  TT_REQUIRE(target - 2 + e->cfg.mel_pos_offset < e->cfg.mel_pos_len, "tt_ar_generate_chunk: %d tokens exceed the mel position table", target);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  if (!first && e->gen_finished) {  // every row already stopped
    *n_total_host = e->gen_done;
    *finished_host = 1;
    return e->sb.leave(us);
  }
  TT_TRY(ar_generate_run(e, B, first != 0, target, ldcodes, sp, codes, n_total_host, finished_host, s));
  return e->sb.leave(us);
}

int tt_ar_stream_latents(tt_ar* e, int B, int n, float* out, void* stream) {
  TT_REQUIRE(e && out, "tt_ar_stream_latents: null argument");
  TT_REQUIRE(e->lat != nullptr, "tt_ar_stream_latents: this handle was created with max_batch %d > 8 (no per-step latent capture)", e->cfg.max_batch);
  TT_REQUIRE(B >= 1 && B <= e->lat_batch && B == e->B && n >= 1 && n <= e->gen_done, "tt_ar_stream_latents: %d x %d latents requested, generation holds %d x %d", B, n, e->B, e->gen_done);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const size_t row = (size_t)e->D * sizeof(float);
  for (int b = 0; b < B; ++b) {
    float* dst = out + (size_t)b * n * e->D;
    // latent 0 is the start-token row of the shared prefill (one copy for every sequence)
    TT_CHECK_HIP(hipMemcpyAsync(dst, e->lat, row, hipMemcpyDeviceToDevice, s));
    if (n > 1)
      TT_CHECK_HIP(hipMemcpy2DAsync(dst + e->D, row, e->lat + ((size_t)e->lat_batch + b) * e->D, (size_t)e->lat_batch * row, row, (size_t)(n - 1),
                                    hipMemcpyDeviceToDevice, s));
  }
  return e->sb.leave(us);
}

int tt_ar_latents(tt_ar* e, const float* emb, int k, int n, float* out, void* stream) {
  TT_REQUIRE(e && emb && out && k >= 1 && n >= 1, "tt_ar_latents: bad arguments");
  TT_REQUIRE(k * n <= e->max_rows && k <= e->cfg.max_batch, "tt_ar_latents: %d x %d rows exceed capacity %d", k, n, e->max_rows);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int D = e->D, M = k * n;
  TT_CHECK_HIP(hipMemcpyAsync(e->x, emb, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
  TT_TRY(gpt_trunk_full(e, k, n, false, s));
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = e->x; a.ldx = D; a.M = M; a.D = D; a.mode = NORM_LAYER;
  a.g1 = e->w.lnf_g; a.b1 = e->w.lnf_b; a.eps1 = 1e-5f;
  a.g2 = e->w.final_norm_g; a.b2 = e->w.final_norm_b; a.eps2 = 1e-5f;
  a.out_f32 = out; a.ldo32 = D;
  a.guard = e->guard;
  TT_TRY(rownorm_launch(e->cfg.dtype, a, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s));
  return e->sb.leave(us);
}

// Operand-overflow guard (fp16 operands saturate at 65504): row norms / sampler launches that met a non-finite value since the
// last reset, as of the end of the last finished tt_ar_generate[_chunk] (which synchronise) or tt_ar_latents (after the caller
// synchronised its stream).  reset != 0 clears the counter.  Returns the count (>= 0) or a negative error.
int tt_ar_guard(tt_ar* e, int reset) {
  if (!e) { set_error("tt_ar_guard: null handle"); return -1; }
  const int n = e->guard_host[0];
  if (n > 0) set_error("autoregressive stage: %d kernel(s) met non-finite values (operand overflow in %s)", n, e->cfg.dtype == DT_F16 ? "fp16: use bf16 operands for this stage" : "bf16");
  if (reset && n > 0) {  // (a clean counter needs no device work: this sits at the end of every utterance)
    if (hipMemsetAsync(e->guard, 0, 4 * sizeof(int), e->sb.own) != hipSuccess || hipStreamSynchronize(e->sb.own) != hipSuccess) { set_error("tt_ar_guard: reset failed"); return -2; }
    e->guard_host[0] = 0;
  }
  return n;
}

// Counters for tests / diagnostics: 0 = decode-step graph captures so far, 1 = queue drains of the launch loop (expected 0),
// 2 = kernel launches of one decode step (layers + head + sampler + step counter).
int tt_ar_stat(tt_ar* e, int which) {
  if (!e) { set_error("tt_ar_stat: null handle"); return -1; }
  const int per_step = 7 * e->cfg.layers + 4 + (e->typical ? 1 : 0);
  return which == 0 ? e->captures : which == 1 ? e->drains : which == 2 ? per_step : -1;
}

// Engine options of a handle (defaults in brackets):
//   TT_AR_OPT_LOOKAHEAD   [6]  decode steps the host may run ahead of the device (>= 1)
int tt_ar_set_option(tt_ar* e, int option, int value) {
  TT_REQUIRE(e != nullptr, "tt_ar_set_option: null handle");
  switch (option) {
    case TT_AR_OPT_LOOKAHEAD:
      TT_REQUIRE(value >= 1 && value <= 64, "tt_ar_set_option: lookahead %d outside 1 .. 64", value);
      e->lookahead = value;
      break;
    default: TT_REQUIRE(false, "tt_ar_set_option: unknown option %d", option);
  }
  return 0;
}

}  // extern "C"
