// Stage 2: DiffusionTts denoiser + SpacedDiffusion.p_sample_loop as an MI355X-native engine
// (reference: tortoise/models/diffusion_decoder.py:232-322, tortoise/utils/diffusion.py:312-621).
//
// Layout: activations are token-major [batch][S][C] so every 1x1 conv is a plain GEMM and every k3
// conv a 3-tap conv-GEMM (gemm.hip).  The conditioned and the conditioning-free evaluation of a
// step run as batch rows 0/1 of ONE pass, so each weight matrix streams once per step instead of
// twice (diffusion.py:341-342 calls the model twice).  Everything that depends only on the
// timestep (time_embed MLP and all 16 ResBlock emb_layers) is evaluated for the whole schedule
// in three GEMMs before the loop.  The conditioning_timestep_integrator (3 DiffusionLayers over code_emb,
// diffusion_decoder.py:292-293) depends on (code_emb, t) but NOT on x_t, so it too leaves the sequential
// loop: all N timesteps x both guidance rows are evaluated as ONE batch of 2N samples before the loop
// (GEMMs with M = 2N*S rows instead of 2S: 128x128 tiles on a full grid), stored in the operand type, and
// each step's integrating conv reads its slice as the second half of K (no concat buffer, no copy).  The sampler maths (CFG blend, learned-range variance, x0 clamp,
// posterior mean, noise) is one fused kernel with no host round trip (the reference does a
// .item() per step, diffusion.py:380).  Per-step scalars, the scale/shift rows and the noise slice
// are all indexed by a device-side step counter, so one captured hipGraph replays every step.
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"

using namespace tt;

static_assert(sizeof(PSampleStep) == sizeof(tt_diff_step), "PSampleStep must mirror tt_diff_step");

// Work buffers of one pass over the denoiser's layers.  Two sets: [0] the sampler loop / conditioning (rows of ONE step), [1] the
// conditioning-integrator pre-pass, which runs over whole chunks of the schedule on its own stream WHILE the sampler loop walks the
// steps whose integrator outputs are already there (diff_sample_run).
struct DiffWork {
  float* tmp_a = nullptr;      // [rows][C] f32 scratch
  float* tmp_b = nullptr;
  float* tmp_c = nullptr;
  void* act = nullptr;         // [rows][C] T   GN/SiLU output (GEMM operand)
  void* q = nullptr; void* k = nullptr; void* vt = nullptr; void* att = nullptr;
  float* gn_partial = nullptr;
  float* gn_gemm_part = nullptr;    // statistics emitted by GEMM epilogues: [row_tile][2][C/16][2]; two buffers used in turn - a GEMM that
  float* gn_gemm_part2 = nullptr;   //   normalises its own A rows (gemm_gna.h) READS its producer's partials while it WRITES its own
  float* stats_part = nullptr;      // the buffer that holds the statistics of stats_ptr
  const float* stats_ptr = nullptr; // tensor those statistics describe (the last such GEMM's f32 output)
  int stats_rows = 0;               // its row-tile height
  int stats_seq = 0;
  int rows = 0;
};

struct tt_diff {
  tt_diff_config cfg;
  tt_diff_weights w;
  std::vector<tt_attn_block> latent_attn, attn;
  std::vector<tt_res_block> res;
  int C, H, NR;  // NR = number of ResBlocks (3 + L + 3)
  int es = 2;    // bytes per operand element (4: the fp32 verification mode)
  Arena arena;
  StreamBridge sb;
  int S = 0;        // rows per sample of the current pass (a padded batch: the common padded length)
  int UB = 1;       // utterance capacity of a batched sampling run (cfg.max_batch)
  int U = 1;        // utterances of the current batch (tt_diff_batch_begin); sample index = guidance row * U + utterance
  int Su[16] = {0}; // their valid lengths
  unsigned conditioned = 0;  // bit u: tt_diff_condition_slot ran for utterance u of the current batch
  bool masked = false;       // the kernels of the current pass take per-sample valid lengths (Su) - off while one utterance is conditioned
  int rows_max = 0;
  float* code_emb = nullptr;   // [2][S][C]: row 0 conditioned, row 1 unconditioned embedding broadcast
  DiffWork wk[2];
  hipStream_t pre_stream = nullptr;          // the pre-pass's stream
  hipEvent_t ev_pre_start = nullptr;
  std::vector<hipEvent_t> ev_chunk;          // chunk c of the pre-pass is complete
  int fuse_gn = 1;                           // TT_DIFF_OPT_FUSED_GN: the sampler step's ResBlock 1x1 GEMMs normalise their own A rows (gemm_gna.h)
  int overlap_prepass = 1;                   // 0: pre-pass on the main stream in front of the loop (round 3); 1: chunks on their own stream, all enqueued up front;
                                             // 2: chunk c + 1 enqueued right before the steps of chunk c (A/B: 297.6 / 294.9 / 296.3 ms, profiles/r04_ab_geometry.txt)
  void* cat = nullptr;         // [2S][C] T     inp_block(x) of the current step (left half of the integrating conv's K)
  void* integ_all = nullptr;   // [steps][B][S][C] T  conditioning_timestep_integrator output of every step (right half of K)
  float* rep_in = nullptr;     // [chunk samples][S][C] f32: code_emb rows repeated per timestep (batched integrator input)
  int chunk_rows = 0;          // row capacity of one batched-integrator chunk
  int integ_B = 0, integ_n = 0; // rows per step / steps currently held by integ_all
  void* lat_t = nullptr;       // [M][latent] T
  int* ts_dev = nullptr;       // [steps]
  float* temb_sin = nullptr;   // [steps][C]
  void* temb_t = nullptr;      // [steps][C] T
  void* temb_t2 = nullptr;     // [steps][C] T
  float* temb_mid = nullptr;   // [steps][C]
  float* ss_all = nullptr;     // [steps][NR][2C]
  float* ss_cur = nullptr;     // [NR][2C]: the CURRENT step's rows at a fixed address (staged by slot_advance_launch)
  int n_steps_cur = 1;
  PSampleStep* steps_dev = nullptr;
  int* slot = nullptr;         // device step counter
  float* x = nullptr;          // [S][in]
  void* x_t = nullptr;         // [2][S][in_pad] T
  float* out = nullptr;        // [2][S][out]
  // operand-overflow guard: device counter bumped by the GroupNorm / sampler kernels when they meet a non-finite value; every
  // sampling run ends with a copy into the pinned word tt_diff_guard reads
  int* guard = nullptr;
  int* guard_host = nullptr;
  // The captured sampler step stays on the handle between calls: the caller's noise / output pointers reach the sampler kernel
  // through a device-side table (io_dev, refreshed per call), so the key is the geometry alone (utterances, lengths, guidance rows,
  // steps).  Replaces the per-call capture + instantiate of round 3 (and the destroy-right-after-the-last-launch that went with it).
  const void** io_dev = nullptr;    // [16][2]: {step_noise, mel_out} per utterance
  const void** io_host = nullptr;   // pinned staging
  hipGraph_t step_graph = nullptr;
  hipGraphExec_t step_exec = nullptr;
  std::vector<int> step_key;
  int captures = 0;  // sampler-step captures so far (tt_diff_stat)
  // split sampling: this handle evaluates one denoiser row per step
  hipGraph_t split_graph = nullptr;
  hipGraphExec_t split_exec = nullptr;
  int split_row = -1, split_steps = 0, split_done = 0;
};

// ss: scale / shift rows [2C]; batch row b reads ss + (b / ss_div) * ss_stride (ss_stride 0: one row for the whole batch)
static int run_gn(tt_diff* e, DiffWork& w_, const float* x, int B, int S, const float* g, const float* b, const float* ss, size_t ss_stride, int ss_div,
                  int act, void* out_t, int ldot, float* out_f32, hipStream_t s) {
  GroupNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.B = B; a.S = S; a.C = e->C; a.gamma = g; a.beta = b; a.eps = 1e-5f;
  a.scale_shift = ss; a.ss_batch_stride = ss_stride; a.ss_batch_div = ss_div; a.act = act;
  // per-step scale / shift rows are staged at a fixed address (e->ss_cur): no step-dependent addressing in the kernel
  a.out_t = out_t; a.ldot = ldot; a.out_f32 = out_f32; a.ldo32 = e->C;
  a.partial = w_.gn_partial;
  a.guard = e->guard;
  if (e->masked) {
    a.vperiod = e->U;
    for (int u = 0; u < e->U; ++u) a.vlen[u] = e->Su[u];
  }
  if (x == w_.stats_ptr && w_.stats_seq == S && S >= w_.stats_rows) {
    a.gemm_part = w_.stats_part;
    a.part_rows = w_.stats_rows;
  }
  return groupnorm_launch(e->cfg.dtype, a, s);
}

// EPI_STD GEMM whose f32 output will be group-normalised next: let its epilogue emit the statistics.
static int gemm_with_stats(tt_diff* e, DiffWork& w_, GemmArgs& g, int S, hipStream_t s) {
  const bool fused = (e->C / 32) % 16 == 0 && g.out_f32 != nullptr && g.N == e->C && g.splitk <= 1;
  float* part = w_.stats_part == w_.gn_gemm_part ? w_.gn_gemm_part2 : w_.gn_gemm_part;  // never the buffer a pending reader holds
  if (fused) {
    g.gn_part = part;
    g.gn_seq = S;
    if (e->masked) {
      g.gn_vperiod = e->U;
      for (int u = 0; u < e->U; ++u) g.gn_vlen[u] = e->Su[u];
    }
  }
  TT_TRY(gemm_launch(e->cfg.dtype, EPI_STD, g, s));
  if (fused) {
    w_.stats_ptr = g.out_f32;
    w_.stats_part = part;
    w_.stats_rows = gemm_stat_rows(g, e->cfg.dtype);
    w_.stats_seq = S;
  } else if (g.out_f32 == w_.stats_ptr) {
    w_.stats_ptr = nullptr;  // tensor overwritten without fresh statistics
  }
  return 0;
}

// AttentionBlock (arch_util.py:80-123): out = in + proj(attn(qkv(GN(in))))
static int run_attn_block(tt_diff* e, DiffWork& w_, const tt_attn_block& w, const float* in, int B, int S, float* out_f32, void* out_t, int ldot,
                          hipStream_t s) {
  const int C = e->C, H = e->H, dt = e->cfg.dtype, M = B * S, n_pad = round_up(S, 32);
  GemmArgs g = gemm_args(w_.act, C, w.w_qkv, C, M, 3 * C, C);
  g.bias = w.b_qkv; g.seq_len = S; g.dmodel = C; g.heads = H; g.q = w_.q; g.k = w_.k; g.vt = w_.vt; g.seq_pad = n_pad;
  g.q_scale = 0.125f;  // (q * 64^-1/4) . (k * 64^-1/4)  ==  (q/8) . k   (arch_util.py:64-67)
  TT_TRY(run_gn(e, w_, in, B, S, w.norm_g, w.norm_b, nullptr, 0, 1, ACT_NONE, w_.act, C, nullptr, s));
  TT_TRY(gemm_launch(dt, EPI_QKV_HEADS, g, s));
  FlashArgs f;
  memset(&f, 0, sizeof(f));
  f.q = w_.q; f.k = w_.k; f.vt = w_.vt; f.out = w_.att; f.ldo = C; f.BH = B * H; f.heads = H; f.n = S; f.n_pad = n_pad;
  f.relpos = w.relpos;
  if (e->masked) {
    f.nv_period = e->U;
    for (int u = 0; u < e->U; ++u) f.nv[u] = e->Su[u];
  }
  TT_TRY(flash_attention_launch(dt, f, s));
  g = gemm_args(w_.att, C, w.w_proj, C, M, C, C);
  g.bias = w.b_proj; g.res = in; g.ldres = C; g.out_f32 = out_f32; g.ldo32 = C; g.out_t = out_t; g.ldot = ldot;
  return gemm_with_stats(e, w_, g, S, s);
}

// ResBlock (diffusion_decoder.py:60-120, use_scale_shift_norm, efficient_config, kernel 3).
static int run_res_block(tt_diff* e, DiffWork& w_, const tt_res_block& w, const float* ss, size_t ss_stride, int ss_div, const float* in, int B, int S,
                         float* out_f32, hipStream_t s) {
  const int C = e->C, dt = e->cfg.dtype, M = B * S;
  GemmArgs g = gemm_args(w_.act, C, w.w_in, C, M, C, C);
  g.bias = w.b_in; g.out_f32 = w_.tmp_c; g.ldo32 = C;
  bool fused_gn = false;
  if (e->fuse_gn && !e->masked && in == w_.stats_ptr && w_.stats_seq == S && S >= w_.stats_rows) {
    // in_layers as ONE launch: GroupNorm32 + SiLU applied on the 1x1 conv's A path (gemm_gna.h), statistics for out_layers' norm in its epilogue
    GemmGnArgs n;
    memset(&n, 0, sizeof(n));
    n.gamma = w.gn1_g; n.beta = w.gn1_b; n.gemm_part = w_.stats_part; n.part_rows = w_.stats_rows; n.S = S; n.eps = 1e-5f; n.act = ACT_SILU;
    n.guard = e->guard;
    GemmArgs gf = g;
    gf.A = in; gf.lda = C;
    float* part = w_.stats_part == w_.gn_gemm_part ? w_.gn_gemm_part2 : w_.gn_gemm_part;
    gf.gn_part = part; gf.gn_seq = S;
    if (gemm_gna_supported(dt, EPI_STD, gf, n)) {
      TT_TRY(gemm_gna_launch(dt, EPI_STD, gf, n, s));
      w_.stats_ptr = gf.out_f32; w_.stats_part = part; w_.stats_rows = gemm_gna_stat_rows(); w_.stats_seq = S;
      fused_gn = true;
    }
  }
  if (!fused_gn) {
    TT_TRY(run_gn(e, w_, in, B, S, w.gn1_g, w.gn1_b, nullptr, 0, 1, ACT_SILU, w_.act, C, nullptr, s));
    TT_TRY(gemm_with_stats(e, w_, g, S, s));
  }
  TT_TRY(run_gn(e, w_, w_.tmp_c, B, S, w.gn2_g, w.gn2_b, ss, ss_stride, ss_div, ACT_SILU, w_.act, C, nullptr, s));
  g = gemm_args(w_.act, C, w.w_out, 3 * C, M, C, 3 * C);
  g.taps = 3; g.seq_len = S; g.bias = w.b_out; g.res = in; g.ldres = C; g.out_f32 = out_f32; g.ldo32 = C;
  return gemm_with_stats(e, w_, g, S, s);
}

// conditioning_timestep_integrator for steps [c0, c0 + ns) x B guidance rows starting at conditioning row `row0`, batched over the
// timesteps: sample (j, r) = (step j, row r) is one batch row of the three DiffusionLayers, with its own scale / shift rows
// ss_all[j] (GroupNorm and attention are per sample, so this is exactly the per-step evaluation, reordered).  Result ->
// integ_all[j][r] in the operand type.  One chunk = at most w_.rows rows of f32 work buffers.
static int diff_integrator_chunk(tt_diff* e, DiffWork& w_, int c0, int ns, int B, int row0, hipStream_t s) {
  const int C = e->C, S = e->S;
  const size_t ss_row = (size_t)e->NR * 2 * C;
  const int nb = ns * B;
  w_.stats_ptr = nullptr;
  TT_TRY(repeat_rows_launch(e->code_emb + (size_t)row0 * S * C, e->rep_in, B * S, ns, C, s));
  const float* cur = e->rep_in;
  for (int i = 0; i < 3; ++i) {
    TT_TRY(run_res_block(e, w_, e->res[i], e->ss_all + (size_t)c0 * ss_row + (size_t)i * 2 * C, ss_row, B, cur, nb, S, w_.tmp_a, s));
    if (i < 2) {
      TT_TRY(run_attn_block(e, w_, e->attn[i], w_.tmp_a, nb, S, w_.tmp_b, nullptr, 0, s));
      cur = w_.tmp_b;
    } else {  // the last layer's output is only ever a GEMM operand: store it in the operand type, per step
      TT_TRY(run_attn_block(e, w_, e->attn[i], w_.tmp_a, nb, S, nullptr, offset_t(e->integ_all, (size_t)c0 * B * S * C, e->es), C, s));
    }
  }
  w_.stats_ptr = nullptr;
  return 0;
}
static inline int diff_steps_per_chunk(const tt_diff* e, int B) { return std::max(1, e->chunk_rows / (B * e->S)); }

// The whole schedule's integrator outputs, chunk after chunk, on stream s (callers that need them all before they go on:
// tt_diff_forward, the split tail).
static int diff_integrator_all(tt_diff* e, int n, int B, int row0, hipStream_t s) {
  const int per = diff_steps_per_chunk(e, B);
  for (int c0 = 0; c0 < n; c0 += per) TT_TRY(diff_integrator_chunk(e, e->wk[1], c0, std::min(per, n - c0), B, row0, s));
  e->integ_B = B;
  e->integ_n = n;
  return 0;
}

// One denoiser evaluation on B batch rows (B = 2: conditioned + unconditioned) for the schedule slot *e->slot; the
// integrator output of that slot must already be in integ_all (diff_integrator_all with the same B / row0).
static int diff_forward(tt_diff* e, int B, hipStream_t s) {
  const int C = e->C, S = e->S, dt = e->cfg.dtype, M = B * S, L = e->cfg.num_layers;
  TT_REQUIRE(e->integ_B == B, "diffusion: the integrator pre-pass holds %d rows per step, this step evaluates %d", e->integ_B, B);
  const float* ss = e->ss_cur;  // the current step's [NR][2C] rows (staged by slot_advance_launch / diff_prepare_timesteps)
  DiffWork& w_ = e->wk[0];
  w_.stats_ptr = nullptr;       // no epilogue statistics are valid at the start of a pass
  // inp_block (k3, in_pad -> C): left half of the integrating conv's K
  GemmArgs g = gemm_args(e->x_t, e->cfg.in_pad, e->w.w_inp, 3 * e->cfg.in_pad, M, C, 3 * e->cfg.in_pad);
  g.taps = 3; g.seq_len = S; g.bias = e->w.b_inp; g.out_t = e->cat; g.ldot = C;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  // integrating_conv over [inp_block(x) | integrator(code_emb, t)]: the right half comes straight from this slot's slice
  g = gemm_args(e->cat, C, e->w.w_integ, 2 * C, M, C, 2 * C);
  g.A2 = e->integ_all; g.lda2 = C; g.k_split = C; g.a2_slot = e->slot; g.a2_slot_stride = (size_t)B * S * C;
  g.bias = e->w.b_integ; g.out_f32 = w_.tmp_a; g.ldo32 = C;
  TT_TRY(gemm_with_stats(e, w_, g, S, s));
  float* hcur = w_.tmp_a;
  float* hoth = w_.tmp_b;
  for (int i = 0; i < L; ++i) {
    TT_TRY(run_res_block(e, w_, e->res[3 + i], ss + (size_t)(3 + i) * 2 * C, 0, 1, hcur, B, S, hoth, s));
    TT_TRY(run_attn_block(e, w_, e->attn[3 + i], hoth, B, S, hcur, nullptr, 0, s));
  }
  for (int i = 0; i < 3; ++i) {
    TT_TRY(run_res_block(e, w_, e->res[3 + L + i], ss + (size_t)(3 + L + i) * 2 * C, 0, 1, hcur, B, S, hoth, s));
    float* t = hcur; hcur = hoth; hoth = t;
  }
  TT_TRY(run_gn(e, w_, hcur, B, S, e->w.out_gn_g, e->w.out_gn_b, nullptr, 0, 1, ACT_SILU, w_.act, C, nullptr, s));
  g = gemm_args(w_.act, C, e->w.w_final, 3 * C, M, e->cfg.out_channels, 3 * C);
  g.taps = 3; g.seq_len = S; g.bias = e->w.b_final; g.out_f32 = e->out; g.ldo32 = e->cfg.out_channels;
  return gemm_launch(dt, EPI_STD, g, s);
}

__global__ void timestep_embedding_kernel(const int* ts, float* out, int n, int dim) {
  // diffusion_decoder.py:21-39: [cos(t * f_j) | sin(t * f_j)], f_j = exp(-ln(10000) * j / half)
  const int half = dim / 2;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n * half; f += gridDim.x * blockDim.x) {
    const int i = f / half, j = f % half;
    const float freq = expf(-logf(10000.f) * (float)j / (float)half);
    const float a = (float)ts[i] * freq;
    out[(size_t)i * dim + j] = cosf(a);
    out[(size_t)i * dim + half + j] = sinf(a);
  }
}

// time_embed MLP + every ResBlock's emb_layers for the n timesteps in e->ts_dev -> ss_all[0..n)
static int diff_prepare_timesteps(tt_diff* e, int n, hipStream_t s) {
  const int C = e->C, dt = e->cfg.dtype;
  timestep_embedding_kernel<<<std::min(cdiv(n * C / 2, 256), 1024), 256, 0, s>>>(e->ts_dev, e->temb_sin, n, C);
  TT_CHECK_HIP(hipGetLastError());
  TT_TRY(cast_pad_launch(dt, e->temb_sin, C, e->temb_t, C, n, C, C, s));
  GemmArgs g = gemm_args(e->temb_t, C, e->w.w_time1, C, n, C, C);
  g.bias = e->w.b_time1; g.act = ACT_SILU; g.out_t = e->temb_t2; g.ldot = C;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  g = gemm_args(e->temb_t2, C, e->w.w_time2, C, n, C, C);
  g.bias = e->w.b_time2; g.out_f32 = e->temb_mid; g.ldo32 = C;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  TT_TRY(silu_cast_launch(dt, e->temb_mid, e->temb_t, n * C, s));
  const int N = e->NR * 2 * C;
  g = gemm_args(e->temb_t, C, e->w.w_emb_all, C, n, N, C);
  g.bias = e->w.b_emb_all; g.out_f32 = e->ss_all; g.ldo32 = N;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  e->n_steps_cur = n;
  TT_CHECK_HIP(hipMemcpyAsync(e->ss_cur, e->ss_all, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, s));  // step 0
  return 0;
}

static void diff_drop_step_graph(tt_diff* e) {
  if (e->step_exec) (void)hipGraphExecDestroy(e->step_exec);
  if (e->step_graph) (void)hipGraphDestroy(e->step_graph);
  e->step_exec = nullptr;
  e->step_graph = nullptr;
  e->step_key.clear();
}

static void split_release(tt_diff* e) {
  if (e->split_exec) (void)hipGraphExecDestroy(e->split_exec);
  if (e->split_graph) (void)hipGraphDestroy(e->split_graph);
  e->split_exec = nullptr;
  e->split_graph = nullptr;
  e->split_row = -1;
  e->split_steps = e->split_done = 0;
}

extern "C" {

int tt_diff_create(const tt_diff_config* cfg, const tt_diff_weights* w, tt_diff** out) {
  TT_REQUIRE(cfg && w && out, "tt_diff_create: null argument");
  TT_REQUIRE(cfg->heads * 64 == cfg->channels, "tt_diff_create: head_dim must be 64");
  TT_REQUIRE(cfg->in_pad % 64 == 0 && cfg->in_pad >= cfg->in_channels && cfg->latent_channels % 64 == 0, "tt_diff_create: in_pad/latent must be multiples of 64");
  TT_REQUIRE(cfg->max_steps >= 1 && cfg->max_seq >= 1 && cfg->max_codes >= 1, "tt_diff_create: bad capacity");
  TT_REQUIRE(cfg->max_batch >= 0 && cfg->max_batch <= 16, "tt_diff_create: max_batch %d outside 0 .. 16", cfg->max_batch);
  tt_diff* e = new tt_diff();
  e->cfg = *cfg;
  e->UB = cfg->max_batch > 1 ? cfg->max_batch : 1;
  e->w = *w;
  e->C = cfg->channels; e->H = cfg->heads; e->NR = 3 + cfg->num_layers + 3;
  e->es = dtype_bytes(cfg->dtype);
  const size_t es = e->es;
  e->latent_attn.assign(w->latent_attn_host, w->latent_attn_host + 4);
  e->attn.assign(w->attn_host, w->attn_host + 3 + cfg->num_layers);
  e->res.assign(w->res_host, w->res_host + e->NR);
  const int C = e->C;
  // batched integrator chunk: as many (step, row) samples as fit ~32k rows, at least one step's two rows
  e->chunk_rows = std::max(2 * e->UB * cfg->max_seq, std::min(32768, cfg->max_steps * 2 * cfg->max_seq));
  e->rows_max = std::max(e->chunk_rows, cfg->max_codes);
  const size_t rows = (size_t)e->rows_max + 64;
  int rc = e->sb.init();
  if (!rc && (hipStreamCreateWithFlags(&e->pre_stream, hipStreamNonBlocking) != hipSuccess ||
              hipEventCreateWithFlags(&e->ev_pre_start, hipEventDisableTiming) != hipSuccess)) { set_error("tt_diff_create: stream / event creation failed"); rc = -2; }
  const size_t B2 = (size_t)2 * e->UB;  // samples of one denoiser pass: (conditioned, conditioning-free) x utterances
  if (!rc) rc = e->arena.alloc_t(&e->code_emb, B2 * cfg->max_seq * C);
  // work set 0: one step's rows (guidance rows x utterances x max_seq) or one conditioning pass (max_codes rows); set 1: a chunk of the pre-pass
  const size_t rows_of[2] = {(size_t)std::max((int)(B2 * cfg->max_seq), cfg->max_codes) + 64, rows};
  for (int i = 0; i < 2 && !rc; ++i) {
    DiffWork& w_ = e->wk[i];
    const size_t r = rows_of[i];
    w_.rows = (int)r;
    rc = e->arena.alloc_t(&w_.tmp_a, r * C);
    if (!rc) rc = e->arena.alloc_t(&w_.tmp_b, r * C);
    if (!rc) rc = e->arena.alloc_t(&w_.tmp_c, r * C);
    if (!rc) rc = e->arena.alloc(&w_.act, r * C * es);
    if (!rc) rc = e->arena.alloc(&w_.q, r * C * es);
    if (!rc) rc = e->arena.alloc(&w_.k, r * C * es);
    if (!rc) rc = e->arena.alloc(&w_.vt, (size_t)C * (2 * r + 64) * es);  // per sample C x round_up(S, 32) keys: <= 2x the rows for short sequences
    if (!rc) rc = e->arena.alloc(&w_.att, r * C * es);
    if (!rc) rc = e->arena.alloc_t(&w_.gn_partial, (r / 16 + r + 64) * 64);  // [samples][row chunks >= 16 rows][32][2], worst case one-row samples
    if (!rc) rc = e->arena.alloc_t(&w_.gn_gemm_part, (r / 32 + 2) * 2 * (C / 16) * 2 + 64);
    if (!rc) rc = e->arena.alloc_t(&w_.gn_gemm_part2, (r / 32 + 2) * 2 * (C / 16) * 2 + 64);
  }
  if (!rc) rc = e->arena.alloc_t(&e->rep_in, rows * C);
  if (!rc) rc = e->arena.alloc(&e->cat, (B2 * cfg->max_seq + 64) * C * es);
  if (!rc) rc = e->arena.alloc(&e->integ_all, ((size_t)cfg->max_steps * B2 * cfg->max_seq + 64) * C * es, false);
  if (!rc) rc = e->arena.alloc(&e->lat_t, ((size_t)cfg->max_codes + 8) * cfg->latent_channels * es);
  if (!rc) rc = e->arena.alloc_t(&e->ts_dev, cfg->max_steps);
  if (!rc) rc = e->arena.alloc_t(&e->temb_sin, (size_t)cfg->max_steps * C);
  if (!rc) rc = e->arena.alloc(&e->temb_t, (size_t)cfg->max_steps * C * es);
  if (!rc) rc = e->arena.alloc(&e->temb_t2, (size_t)cfg->max_steps * C * es);
  if (!rc) rc = e->arena.alloc_t(&e->temb_mid, (size_t)cfg->max_steps * C);
  if (!rc) rc = e->arena.alloc_t(&e->ss_all, (size_t)cfg->max_steps * e->NR * 2 * C);
  if (!rc) rc = e->arena.alloc_t(&e->ss_cur, (size_t)e->NR * 2 * C);
  if (!rc) rc = e->arena.alloc_t(&e->steps_dev, cfg->max_steps);
  if (!rc) rc = e->arena.alloc_t(&e->slot, 4);
  if (!rc) rc = e->arena.alloc_t(&e->x, (size_t)e->UB * cfg->max_seq * cfg->in_channels);
  if (!rc) rc = e->arena.alloc(&e->x_t, (B2 * cfg->max_seq + 8) * cfg->in_pad * es);
  if (!rc) rc = e->arena.alloc_t(&e->out, B2 * cfg->max_seq * cfg->out_channels);
  if (!rc) rc = e->arena.alloc_t(&e->guard, 4);
  if (!rc) rc = e->arena.alloc_t(&e->io_dev, 32);
  if (!rc && (hipHostMalloc((void**)&e->guard_host, 4 * sizeof(int)) != hipSuccess || hipHostMalloc((void**)&e->io_host, 32 * sizeof(void*)) != hipSuccess)) {
    set_error("tt_diff_create: hipHostMalloc failed");
    rc = -2;
  }
  if (!rc) e->guard_host[0] = 0;
  if (rc) {
    tt_diff_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_diff_destroy(tt_diff* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  split_release(e);
  diff_drop_step_graph(e);
  if (e->guard_host) (void)hipHostFree(e->guard_host);
  if (e->io_host) (void)hipHostFree((void*)e->io_host);
  for (hipEvent_t ev : e->ev_chunk)
    if (ev) (void)hipEventDestroy(ev);
  if (e->ev_pre_start) (void)hipEventDestroy(e->ev_pre_start);
  if (e->pre_stream) (void)hipStreamDestroy(e->pre_stream);
  e->arena.release();
  e->sb.destroy();
  delete e;
}

// DiffusionTts.timestep_independent for ONE utterance: its S conditioned rows -> dst_cond, the unconditioned embedding
// broadcast -> dst_uncond (both [S][C] f32).  Runs on the M code rows alone (no padding, no masks).
static int diff_condition_into(tt_diff* e, const float* latents, int M, const float* cond, const int* interp_idx, int S, float* dst_cond,
                               float* dst_uncond, hipStream_t s) {
  const int C = e->C, dt = e->cfg.dtype, LC = e->cfg.latent_channels;
  const bool masked = e->masked;
  DiffWork& w_ = e->wk[0];
  e->masked = false;
  w_.stats_ptr = nullptr;
  TT_TRY(cast_pad_launch(dt, latents, LC, e->lat_t, LC, M, LC, LC, s));
  GemmArgs g = gemm_args(e->lat_t, LC, e->w.w_latent_conv, 3 * LC, M, C, 3 * LC);
  g.taps = 3; g.seq_len = M; g.bias = e->w.b_latent_conv; g.out_f32 = w_.tmp_a; g.ldo32 = C;
  TT_TRY(gemm_with_stats(e, w_, g, M, s));
  float* cur = w_.tmp_a;
  float* oth = w_.tmp_b;
  for (int i = 0; i < 4; ++i) {
    TT_TRY(run_attn_block(e, w_, e->latent_attn[i], cur, 1, M, oth, nullptr, 0, s));
    float* t = cur; cur = oth; oth = t;
  }
  // code_norm(h) * (1 + cond_scale) + cond_shift   (diffusion_decoder.py:249-250)
  TT_TRY(run_gn(e, w_, cur, 1, M, e->w.code_norm_g, e->w.code_norm_b, cond, 0, 1, ACT_NONE, nullptr, 0, oth, s));
  TT_TRY(gather_rows_launch(oth, interp_idx, dst_cond, S, C, s));        // F.interpolate(nearest)
  TT_TRY(broadcast_rows_launch(e->w.uncond_emb, dst_uncond, S, C, s));   // unconditioned_embedding.repeat
  w_.stats_ptr = nullptr;
  e->masked = masked;
  return 0;
}

int tt_diff_condition(tt_diff* e, const float* latents, int M, const float* cond, const int* interp_idx, int S, void* stream) {
  TT_REQUIRE(e && latents && cond && interp_idx, "tt_diff_condition: null argument");
  TT_REQUIRE(M >= 1 && M <= e->cfg.max_codes && S >= 1 && S <= e->cfg.max_seq, "tt_diff_condition: M=%d S=%d exceed capacity (%d, %d)", M, S, e->cfg.max_codes, e->cfg.max_seq);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  e->S = S;
  e->U = 1; e->Su[0] = S; e->masked = false; e->conditioned = 1;
  TT_TRY(diff_condition_into(e, latents, M, cond, interp_idx, S, e->code_emb, e->code_emb + (size_t)S * e->C, s));
  return e->sb.leave(us);
}

int tt_diff_batch_begin(tt_diff* e, int U, int S_pad, void* stream) {
  TT_REQUIRE(e != nullptr, "tt_diff_batch_begin: null handle");
  TT_REQUIRE(U >= 1 && U <= e->UB, "tt_diff_batch_begin: %d utterances exceed this handle's capacity (%d; tt_diff_config.max_batch)", U, e->UB);
  TT_REQUIRE(S_pad >= 1 && S_pad <= e->cfg.max_seq, "tt_diff_batch_begin: padded length %d exceeds capacity %d", S_pad, e->cfg.max_seq);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  e->U = U; e->S = S_pad; e->conditioned = 0; e->masked = false;
  for (int u = 0; u < 16; ++u) e->Su[u] = 0;
  // rows past an utterance's own length stay zero for the whole run: they are the zero padding its convolutions read
  TT_CHECK_HIP(hipMemsetAsync(e->code_emb, 0, (size_t)2 * U * S_pad * e->C * sizeof(float), s));
  TT_CHECK_HIP(hipMemsetAsync(e->x_t, 0, (size_t)2 * U * S_pad * e->cfg.in_pad * e->es, s));
  TT_CHECK_HIP(hipMemsetAsync(e->x, 0, (size_t)U * S_pad * e->cfg.in_channels * sizeof(float), s));
  return e->sb.leave(us);
}

int tt_diff_condition_slot(tt_diff* e, int u, const float* latents, int M, const float* cond, const int* interp_idx, int S, void* stream) {
  TT_REQUIRE(e && latents && cond && interp_idx, "tt_diff_condition_slot: null argument");
  TT_REQUIRE(u >= 0 && u < e->U, "tt_diff_condition_slot: utterance %d outside the batch of %d (tt_diff_batch_begin)", u, e->U);
  TT_REQUIRE(M >= 1 && M <= e->cfg.max_codes && S >= 1 && S <= e->S, "tt_diff_condition_slot: M=%d S=%d exceed capacity (%d codes, padded length %d)", M, S, e->cfg.max_codes, e->S);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const size_t slot = (size_t)e->S * e->C;
  e->Su[u] = S;
  e->conditioned |= 1u << u;
  TT_TRY(diff_condition_into(e, latents, M, cond, interp_idx, S, e->code_emb + (size_t)u * slot, e->code_emb + (size_t)(e->U + u) * slot, s));
  return e->sb.leave(us);
}

int tt_diff_get_code_emb(tt_diff* e, float* dst, void* stream) {
  TT_REQUIRE(e && dst && e->S > 0, "tt_diff_get_code_emb: no conditioning");
  TT_REQUIRE(e->U == 1 && e->conditioned == 1u, "tt_diff_get_code_emb: the handle holds a batch of %d utterances (tt_diff_batch_begin); call tt_diff_condition first", e->U);
  hipStream_t us = (hipStream_t)stream;
  TT_TRY(e->sb.enter(us));
  TT_CHECK_HIP(hipMemcpyAsync(dst, e->code_emb, (size_t)e->S * e->C * sizeof(float), hipMemcpyDeviceToDevice, e->sb.own));
  return e->sb.leave(us);
}

int tt_diff_forward(tt_diff* e, const float* x, int timestep, int cond_free, float* out, void* stream) {
  TT_REQUIRE(e && x && out && e->S > 0, "tt_diff_forward: call tt_diff_condition first");
  TT_REQUIRE(e->U == 1 && e->conditioned == 1u, "tt_diff_forward: the handle holds a batch of %d utterances (tt_diff_batch_begin); call tt_diff_condition first", e->U);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int S = e->S, IC = e->cfg.in_channels, IP = e->cfg.in_pad;
  TT_CHECK_HIP(hipMemcpyAsync(e->ts_dev, &timestep, sizeof(int), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipStreamSynchronize(s));  // `timestep` lives on the caller's stack
  TT_CHECK_HIP(hipMemsetAsync(e->slot, 0, sizeof(int), s));
  TT_TRY(diff_prepare_timesteps(e, 1, s));
  TT_TRY(cast_pad_launch(e->cfg.dtype, x, IC, e->x_t, IP, S, IC, IP, s));
  TT_TRY(cast_pad_launch(e->cfg.dtype, x, IC, offset_t(e->x_t, (size_t)S * IP, e->es), IP, S, IC, IP, s));
  const int B = cond_free ? 2 : 1;
  TT_TRY(diff_integrator_all(e, 1, B, 0, s));
  TT_TRY(diff_forward(e, B, s));
  TT_CHECK_HIP(hipMemcpyAsync(out, e->out, (size_t)B * S * e->cfg.out_channels * sizeof(float), hipMemcpyDeviceToDevice, s));
  return e->sb.leave(us);
}

// p_sample_loop for the U utterances of the current batch (U = 1: the plain single-utterance run).  All of them walk the same
// schedule; utterance u has Su[u] positions inside the padded length e->S.
static int diff_sample_run(tt_diff* e, const float* const* x_T, const float* const* step_noise, const tt_diff_step* steps_host, int n_steps,
                           int cond_free, float* const* mel_out, hipStream_t s) {
  const int S = e->S, U = e->U, IC = e->cfg.in_channels, IP = e->cfg.in_pad, dt = e->cfg.dtype;
  std::vector<int> ts(n_steps);
  for (int i = 0; i < n_steps; ++i) ts[i] = steps_host[i].timestep;
  TT_CHECK_HIP(hipMemcpyAsync(e->ts_dev, ts.data(), n_steps * sizeof(int), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->steps_dev, steps_host, n_steps * sizeof(tt_diff_step), hipMemcpyHostToDevice, s));
  for (int u = 0; u < U; ++u) {  // this call's noise / output pointers: data for the kept sampler-step graph
    e->io_host[2 * u] = step_noise[u];
    e->io_host[2 * u + 1] = mel_out[u];
  }
  TT_CHECK_HIP(hipMemcpyAsync(e->io_dev, e->io_host, 32 * sizeof(void*), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipStreamSynchronize(s));  // host staging buffers may go away (and io_host may be rewritten by the next call)
  TT_CHECK_HIP(hipMemsetAsync(e->slot, 0, sizeof(int), s));
  TT_TRY(diff_prepare_timesteps(e, n_steps, s));
  const int R = cond_free ? 2 : 1, B = R * U;
  for (int u = 0; u < U; ++u) {
    float* xu = e->x + (size_t)u * S * IC;
    TT_TRY(transpose_launch(x_T[u], xu, IC, e->Su[u], s));  // [C][S_u] -> [S_u][C]
    for (int r = 0; r < R; ++r) TT_TRY(cast_pad_launch(dt, xu, IC, offset_t(e->x_t, (size_t)(r * U + u) * S * IP, e->es), IP, e->Su[u], IC, IP, s));
  }
  e->masked = U > 1;
  // Every step's conditioning integrator, batched over the schedule in chunks of `per` steps.  The chunks run on the pre-pass stream
  // with their own work buffers; the sampler loop below starts as soon as chunk 0 is there and waits, in front of the first step of
  // every later chunk, for that chunk's event: the large MFMA-bound GEMMs of the pre-pass (M = chunk rows) fill the CUs the loop's
  // latency-bound launches (M = 2 S rows) leave idle.  overlap_prepass == 0: the round-3 order (whole pre-pass first, one stream).
  const int per = diff_steps_per_chunk(e, B), nchunks = cdiv(n_steps, per);
  const bool overlap = e->overlap_prepass != 0 && nchunks > 1;
  int rc = 0;
  hipStream_t ps = overlap ? e->pre_stream : s;
  while (!rc && (int)e->ev_chunk.size() < nchunks) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("tt_diff_sample: event creation failed"); rc = -2; }
    else e->ev_chunk.push_back(ev);
  }
  if (!rc && overlap) {
    if (hipEventRecord(e->ev_pre_start, s) != hipSuccess || hipStreamWaitEvent(ps, e->ev_pre_start, 0) != hipSuccess) { set_error("tt_diff_sample: fork failed"); rc = -2; }
  }
  // chunk c is enqueued (on the pre-pass stream) right before the loop steps of chunk c - 1 are launched, so the main stream's first
  // step is in the queue after ONE chunk's worth of host launches, not after all of them (eager launches: ~0.4 ms of host time per chunk)
  int chunks_enqueued = 0;
  auto enqueue_chunk = [&]() -> int {
    const int c = chunks_enqueued++;
    TT_TRY(diff_integrator_chunk(e, e->wk[1], c * per, std::min(per, n_steps - c * per), B, 0, ps));
    if (overlap) TT_CHECK_HIP(hipEventRecord(e->ev_chunk[c], ps));
    return 0;
  };
  if (!rc) rc = enqueue_chunk();
  if (!overlap || e->overlap_prepass == 1)  // (1: every chunk goes out before the first step; 2: chunk c + 1 right before the steps of chunk c)
    while (!rc && chunks_enqueued < nchunks) rc = enqueue_chunk();
  e->integ_B = B;
  e->integ_n = n_steps;
  auto chunk_gate = [&](int step) -> int {  // in front of step `step`: its integrator slice must be there
    if (overlap && step % per == 0) {
      if (chunks_enqueued < nchunks) TT_TRY(enqueue_chunk());  // the NEXT chunk goes out before this chunk's steps
      TT_CHECK_HIP(hipStreamWaitEvent(s, e->ev_chunk[step / per], 0));
    }
    return 0;
  };
  std::vector<PSampleArgs> pa(U);
  for (int u = 0; u < U && !rc; ++u) {
    PSampleArgs& p = pa[u];
    memset(&p, 0, sizeof(p));
    p.steps = e->steps_dev; p.slot = e->slot; p.x = e->x + (size_t)u * S * IC; p.x_t = offset_t(e->x_t, (size_t)u * S * IP, e->es); p.cpad = IP;
    p.out = e->out + (size_t)u * S * e->cfg.out_channels;
    p.has_uncond = cond_free ? 1 : 0; p.S = e->Su[u]; p.C = IC;
    p.io = e->io_dev + 2 * u;  // {step_noise[u], mel_out[u]}: data of this call, not of the captured step
    p.guard = e->guard;
    p.ld_rows = U * S;
    p.mel_scale = 2.3143386840820312f - (-11.512925148010254f);
    p.mel_shift = -11.512925148010254f;
  }
  auto one_step = [&]() -> int {
    TT_TRY(diff_forward(e, B, s));
    for (int u = 0; u < U; ++u) TT_TRY(psample_launch(dt, pa[u], s));
    return slot_advance_launch(e->slot, e->ss_all, e->ss_cur, e->NR * 2 * e->C, e->n_steps_cur - 1, s);
  };
  if (!rc && graphs_enabled() && n_steps > 2) {
    // everything the captured step bakes in that a later call could change
    std::vector<int> key = {U, S, R, n_steps, dt, g_prof_on ? 1 : 0};
    for (int u = 0; u < 16; ++u) key.push_back(u < U ? e->Su[u] : 0);
    if (!e->step_exec || key != e->step_key) {
      diff_drop_step_graph(e);
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hipError_t ce = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (ce != hipSuccess) { set_error("tt_diff_sample: capture failed: %s", hipGetErrorString(ce)); rc = -2; }
      if (!rc) {
        rc = one_step();
        ce = hipStreamEndCapture(s, &graph);
        if (!rc && ce != hipSuccess) { set_error("tt_diff_sample: capture failed: %s", hipGetErrorString(ce)); rc = -2; }
      }
      if (!rc) {
        ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ce != hipSuccess) { set_error("tt_diff_sample: instantiate failed: %s", hipGetErrorString(ce)); rc = -2; }
      }
      if (rc) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
      } else {
        e->step_graph = graph;
        e->step_exec = exec;
        e->step_key.swap(key);
        e->captures += 1;
      }
    }
    for (int i = 0; i < n_steps && !rc; ++i) {
      rc = chunk_gate(i);
      if (rc) break;
      hipError_t ce = hipGraphLaunch(e->step_exec, s);
      if (ce != hipSuccess) { set_error("tt_diff_sample: hipGraphLaunch: %s", hipGetErrorString(ce)); rc = -2; }
    }
    if (rc) {
      (void)hipStreamSynchronize(s);
      diff_drop_step_graph(e);
    }
  } else {
    for (int i = 0; i < n_steps && !rc; ++i) {
      rc = chunk_gate(i);
      if (!rc) rc = one_step();
    }
  }
  if (rc && overlap) (void)hipStreamSynchronize(ps);  // nothing of this run may still be in flight when the caller sees the error
  if (!rc && hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) { set_error("tt_diff_sample: reading the guard failed"); rc = -2; }
  e->masked = false;
  return rc;
}

int tt_diff_sample(tt_diff* e, const float* x_T, const float* step_noise, const tt_diff_step* steps_host, int n_steps, int cond_free,
                   float* mel_out, void* stream) {
  TT_REQUIRE(e && x_T && steps_host && mel_out && e->S > 0, "tt_diff_sample: call tt_diff_condition first");
  TT_REQUIRE(e->U == 1 && e->conditioned == 1u, "tt_diff_sample: the handle holds a batch of %d utterances (tt_diff_sample_batch)", e->U);
  TT_REQUIRE(n_steps >= 1 && n_steps <= e->cfg.max_steps, "tt_diff_sample: %d steps exceed capacity %d", n_steps, e->cfg.max_steps);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  TT_TRY(diff_sample_run(e, &x_T, &step_noise, steps_host, n_steps, cond_free, &mel_out, s));
  return e->sb.leave(us);
}

int tt_diff_sample_batch(tt_diff* e, int U, const float* const* x_T, const float* const* step_noise, const tt_diff_step* steps_host, int n_steps,
                         int cond_free, float* const* mel_out, void* stream) {
  TT_REQUIRE(e && x_T && step_noise && steps_host && mel_out, "tt_diff_sample_batch: null argument");
  TT_REQUIRE(U == e->U && e->conditioned == (U >= 32 ? ~0u : (1u << U) - 1u), "tt_diff_sample_batch: %d utterances, but the batch has %d and conditioning mask %#x (tt_diff_batch_begin / tt_diff_condition_slot)", U, e->U, e->conditioned);
  TT_REQUIRE(n_steps >= 1 && n_steps <= e->cfg.max_steps, "tt_diff_sample_batch: %d steps exceed capacity %d", n_steps, e->cfg.max_steps);
  for (int u = 0; u < U; ++u) TT_REQUIRE(x_T[u] && mel_out[u] && (step_noise[u] || n_steps == 1), "tt_diff_sample_batch: null tensor for utterance %d", u);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  TT_TRY(diff_sample_run(e, x_T, step_noise, steps_host, n_steps, cond_free, mel_out, s));
  return e->sb.leave(us);
}

int tt_diff_split_begin(tt_diff* e, const float* x_T, const tt_diff_step* steps_host, int n_steps, int row, void* stream) {
  TT_REQUIRE(e && x_T && steps_host && e->S > 0, "tt_diff_split_begin: call tt_diff_condition first");
  TT_REQUIRE(n_steps >= 1 && n_steps <= e->cfg.max_steps, "tt_diff_split_begin: %d steps exceed capacity %d", n_steps, e->cfg.max_steps);
  TT_REQUIRE(row == 0 || row == 1, "tt_diff_split_begin: row must be 0 (conditioned) or 1 (conditioning-free)");
  TT_REQUIRE(e->U == 1 && e->conditioned == 1u, "tt_diff_split_begin: the handle holds a batch of %d utterances (tt_diff_batch_begin); call tt_diff_condition first", e->U);
  split_release(e);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int S = e->S, IC = e->cfg.in_channels, IP = e->cfg.in_pad, dt = e->cfg.dtype;
  std::vector<int> ts(n_steps);
  for (int i = 0; i < n_steps; ++i) ts[i] = steps_host[i].timestep;
  TT_CHECK_HIP(hipMemcpyAsync(e->ts_dev, ts.data(), n_steps * sizeof(int), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->steps_dev, steps_host, n_steps * sizeof(tt_diff_step), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipStreamSynchronize(s));  // host staging buffers may go away
  TT_CHECK_HIP(hipMemsetAsync(e->slot, 0, sizeof(int), s));
  TT_TRY(diff_prepare_timesteps(e, n_steps, s));
  TT_TRY(transpose_launch(x_T, e->x, IC, S, s));  // [C][S] -> [S][C]
  TT_TRY(cast_pad_launch(dt, e->x, IC, e->x_t, IP, S, IC, IP, s));
  TT_TRY(cast_pad_launch(dt, e->x, IC, offset_t(e->x_t, (size_t)S * IP, e->es), IP, S, IC, IP, s));
  TT_TRY(diff_integrator_all(e, n_steps, 1, row, s));
  int rc = 0;
  if (graphs_enabled()) {
    TT_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rc = diff_forward(e, 1, s);
    hipError_t ce = hipStreamEndCapture(s, &e->split_graph);
    if (!rc && ce != hipSuccess) { set_error("tt_diff_split_begin: capture failed: %s", hipGetErrorString(ce)); rc = -2; }
    if (!rc) {
      ce = hipGraphInstantiate(&e->split_exec, e->split_graph, nullptr, nullptr, 0);
      if (ce != hipSuccess) { set_error("tt_diff_split_begin: instantiate failed: %s", hipGetErrorString(ce)); rc = -2; }
    }
    if (rc) split_release(e);
  }
  TT_TRY(rc);
  e->split_row = row;
  e->split_steps = n_steps;
  e->split_done = 0;
  return e->sb.leave(us);
}

int tt_diff_split_forward(tt_diff* e, float* out_row, void* stream) {
  TT_REQUIRE(e && out_row && e->split_row >= 0, "tt_diff_split_forward: call tt_diff_split_begin first");
  TT_REQUIRE(e->split_done < e->split_steps, "tt_diff_split_forward: all %d steps already ran", e->split_steps);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  if (e->split_exec) {
    hipError_t ce = hipGraphLaunch(e->split_exec, s);
    if (ce != hipSuccess) { set_error("tt_diff_split_forward: hipGraphLaunch: %s", hipGetErrorString(ce)); return -2; }
  } else {
    TT_TRY(diff_forward(e, 1, s));
  }
  TT_CHECK_HIP(hipMemcpyAsync(out_row, e->out, (size_t)e->S * e->cfg.out_channels * sizeof(float), hipMemcpyDeviceToDevice, s));
  return e->sb.leave(us);
}

int tt_diff_split_update(tt_diff* e, const float* rows, const float* step_noise, float* mel_out, void* stream) {
  TT_REQUIRE(e && rows && mel_out && e->split_row >= 0, "tt_diff_split_update: call tt_diff_split_begin first");
  TT_REQUIRE(e->split_done < e->split_steps, "tt_diff_split_update: all %d steps already ran", e->split_steps);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  PSampleArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.steps = e->steps_dev; pa.slot = e->slot; pa.x = e->x; pa.x_t = e->x_t; pa.cpad = e->cfg.in_pad; pa.out = rows;
  pa.has_uncond = 1; pa.noise = step_noise; pa.S = e->S; pa.C = e->cfg.in_channels;
  pa.mel_out = mel_out;
  pa.guard = e->guard;
  pa.mel_scale = 2.3143386840820312f - (-11.512925148010254f);
  pa.mel_shift = -11.512925148010254f;
  TT_TRY(psample_launch(e->cfg.dtype, pa, s));
  TT_TRY(slot_advance_launch(e->slot, e->ss_all, e->ss_cur, e->NR * 2 * e->C, e->n_steps_cur - 1, s));
  e->split_done += 1;
  return e->sb.leave(us);
}

int tt_diff_split_end(tt_diff* e) {
  TT_REQUIRE(e != nullptr, "tt_diff_split_end: null handle");
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, e->sb.own));
  TT_CHECK_HIP(hipStreamSynchronize(e->sb.own));
  split_release(e);
  return 0;
}

int tt_diff_stat(tt_diff* e, int which) {  // 0: sampler-step graph captures so far (tests: the kept graph is reused)
  if (!e) { set_error("tt_diff_stat: null handle"); return -1; }
  return which == 0 ? e->captures : -1;
}

// TT_DIFF_OPT_OVERLAP_PREPASS [1]: the conditioning-integrator pre-pass runs chunk by chunk on its own stream while the sampler loop
// walks the steps whose chunks are done (0: the whole pre-pass first, on the one stream - the round-3 order; same results either way).
int tt_diff_set_option(tt_diff* e, int option, int value) {
  TT_REQUIRE(e != nullptr, "tt_diff_set_option: null handle");
  TT_REQUIRE(option == TT_DIFF_OPT_OVERLAP_PREPASS || option == TT_DIFF_OPT_FUSED_GN, "tt_diff_set_option: unknown option %d", option);
  if (option == TT_DIFF_OPT_FUSED_GN) {
    if (value != e->fuse_gn) diff_drop_step_graph(e);  // the kept sampler step was captured with the other launch sequence
    TT_REQUIRE(value == 0 || value == 1, "tt_diff_set_option: TT_DIFF_OPT_FUSED_GN takes 0 (stand-alone applies) or 1 (default: ResBlock in_layers fused), got %d", value);
    e->fuse_gn = value;
    return 0;
  }
  e->overlap_prepass = value < 0 ? 0 : value > 2 ? 2 : value;
  return 0;
}

// Operand-overflow guard (fp16 operands saturate at 65504): non-finite values met by the GroupNorm statistics / the sampler since
// the last reset, as of the end of the last finished tt_diff_sample / tt_diff_sample_batch / tt_diff_split_end.  reset != 0 clears it.
int tt_diff_guard(tt_diff* e, int reset) {
  if (!e) { set_error("tt_diff_guard: null handle"); return -1; }
  const int n = e->guard_host[0];
  if (n > 0) set_error("diffusion stage: %d kernel(s) met non-finite values (operand overflow in %s)", n, e->cfg.dtype == DT_F16 ? "fp16: re-run this stage with bf16 operands" : "bf16");
  if (reset && n > 0) {  // (a clean counter needs no device work: this sits at the end of every utterance)
    if (hipMemsetAsync(e->guard, 0, 4 * sizeof(int), e->sb.own) != hipSuccess || hipStreamSynchronize(e->sb.own) != hipSuccess) { set_error("tt_diff_guard: reset failed"); return -2; }
    e->guard_host[0] = 0;
  }
  return n;
}

}  // extern "C"
