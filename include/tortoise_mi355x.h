/* tortoise_mi355x.h — C-ABI of the MI355X (gfx950) Tortoise inference engine.
 *
 * The reference (neonbjb/tortoise-tts) has no FFI: its hot path is five Python call sites inside
 * TextToSpeech.tts() (tortoise/api.py).  Each entry-point group below replaces exactly one of those
 * call sites; the file:line it replaces is cited.  Host code (tortoise_tts_amd/api.py) keeps the
 * reference's Python signature and calls these through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on error; tt_last_error() has the text.
 *     Nothing throws across the boundary.
 *   - all tensor arguments are DEVICE pointers unless the name ends in _host; the caller owns them.
 *     Weight pointers passed to *_create must stay valid until *_destroy (the engine keeps them,
 *     it does not copy weights).
 *   - `stream` is a hipStream_t (0 = default stream).  Calls are asynchronous on that stream unless
 *     stated otherwise; no call allocates device memory after *_create / *_reserve.
 *   - dtype: TT_BF16 or TT_F16 selects the MFMA operand type (weights + GEMM activations; TT_F32: see below);
 *     residual streams, norms, softmax and accumulators are always f32.
 *   - "T" below means that operand type.  Weight matrices are [out_features][in_features]
 *     (K contiguous); conv kernels are [out][tap][in_padded].  tortoise_tts_amd/pack.py produces
 *     these from reference-layout state_dicts.
 */
#ifndef TORTOISE_MI355X_H
#define TORTOISE_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_BF16 0
#define TT_F16 1
/* Verification mode: fp32 GEMM / attention operands and fp32 KV caches in the autoregressive, CLVP, diffusion and vocoder stages
 * (v_mfma_f32_16x16x4_f32 + plain VALU attention, untuned - an order of magnitude slower).  It exists so that tests can hold the
 * engines against the reference's fp32 modules at fp32 tolerances instead of inside bf16 / fp16 operand noise; weight
 * matrices are then f32 in the layouts documented below ("T" = float).  Other stage handles refuse it. */
#define TT_F32 2

const char* tt_last_error(void);
int tt_init(void);     /* once per process, after the HIP device is selected */
int tt_abi_version(void);
size_t tt_struct_size(int which);  /* sizeof of boundary struct #which (order of declaration below) */

/* ============================================================================================
 * Stage 1 — UnifiedVoice / GPT2InferenceModel   (reference: tortoise/models/autoregressive.py)
 * ============================================================================================ */
typedef struct tt_gpt_layer {
  const float* ln1_g; const float* ln1_b;     /* gpt.h.i.ln_1 */
  const void* w_qkv;  const float* b_qkv;     /* T [3D][D]  (HF Conv1D c_attn transposed) */
  const void* w_proj; const float* b_proj;    /* T [D][D] */
  const float* ln2_g; const float* ln2_b;
  const void* w_fc;   const float* b_fc;      /* T [4D][D] */
  const void* w_proj2; const float* b_proj2;  /* T [D][4D] */
} tt_gpt_layer;

typedef struct tt_ar_config {
  int dtype;
  int layers, model_dim, heads;
  int vocab;             /* number_mel_codes (8194) */
  int start_mel_token, stop_mel_token;
  int mel_pos_len;       /* rows of mel_pos_embedding */
  int max_batch;         /* candidates decoded together.  A handle of <= 4 (16-bit operands, one group) is a streaming-size handle: its decode
                          * step runs GEMV-shaped kernels (csrc/gemv.hip) - deterministic and the same for every call on the handle, but another
                          * summation order than larger handles; create handles of >= 5 where codes must not depend on the capacity */
  int max_prefix;        /* max P+1 (conditioning + text + start token) */
  int max_new_tokens;    /* per-sequence KV slots (the handle allocates the next multiple of 8: cache strides next to a multiple of 4 KB
                          * slow the decode attention down by up to 30 %) */
  int max_full_rows;     /* rows of the largest teacher-forced pass (k * (1 + T+2 + M+2)) */
  int mel_pos_offset;    /* mel position row of generated token i (i >= 0; the start token uses row 0) = i + mel_pos_offset:
                          * 2 = TextToSpeech(kv_cache=True): rows 0,2,3,... (autoregressive.py:145-149, attention_mask.shape[1] - mel_len)
                          * 1 = TextToSpeech(kv_cache=False), the reference DEFAULT: rows 0,1,2,... (autoregressive.py:134-144) */
  int max_groups;        /* utterances decoded in ONE batch (tt_ar_prefill_group), 1 .. 16; 0 is read as 1.  max_batch counts the
                          * sequences of all groups together */
} tt_ar_config;

typedef struct tt_ar_weights {
  const tt_gpt_layer* layers_host;  /* HOST array of `layers` entries (device pointers inside) */
  const float* lnf_g; const float* lnf_b;               /* gpt.ln_f */
  const float* final_norm_g; const float* final_norm_b; /* final_norm */
  const void* w_mel_head; const float* b_mel_head;      /* T [vocab][D] */
  const float* mel_emb;   /* f32 [vocab][D]   mel_embedding.weight */
  const float* mel_pos;   /* f32 [mel_pos_len][D]  mel_pos_embedding.emb.weight */
} tt_ar_weights;

typedef struct tt_ar tt_ar;
int tt_ar_create(const tt_ar_config* cfg, const tt_ar_weights* w, tt_ar** out);
void tt_ar_destroy(tt_ar* h);

/* Replaces the first GenerationMixin step of UnifiedVoice.inference_speech
 * (autoregressive.py:538-549 prefix + 134-144 prefill branch, called from api.py:416-424).
 * prefix_emb f32 [P][D] = [cond latent | text_emb + text_pos].  The start-token row is appended
 * here.  All candidates share this prefix, so it is evaluated ONCE (M = P+1 rows) and its K/V are
 * shared by every sequence of the following tt_ar_generate / tt_ar_decode_step calls. */
int tt_ar_prefill(tt_ar* h, const float* prefix_emb, int P, void* stream);
/* Several utterances in one decode batch (long-form reading, tortoise/read.py:66-71 renders its chunks one after the other; 288 GB
 * of HBM hold the KV caches of many): group g of n_groups gets its own prefix (its own text, its own voice), evaluated once like
 * tt_ar_prefill's.  After all n_groups groups have been prefilled, tt_ar_generate with B = n_groups * group_size decodes them
 * together: sequences [g * group_size, (g + 1) * group_size) attend to prefix g, sample their first token from prefix g's logits
 * and draw from Philox streams keyed by (tt_sampling.group_seeds[g] or seed, row_offset + index WITHIN the group) - so each
 * group's codes are bit-identical to decoding it alone.  group_size must be a multiple of 4.  tt_ar_prefill == group 0 of 1. */
int tt_ar_prefill_group(tt_ar* h, int group, int n_groups, const float* prefix_emb, int P, void* stream);

/* Logits of the newest position: f32 [rows][vocab]; rows = 1 after prefill, B after a decode step. */
int tt_ar_get_logits(tt_ar* h, float* dst, int rows, void* stream);

typedef struct tt_sampling {
  float temperature, top_p, repetition_penalty;
  int top_k;                 /* HF GenerationConfig default 50 (api.py never overrides it).  Any value: 1 .. 256 run the fast sampler,
                              * larger ones a full-sort sampler; <= 0 or >= vocab means no top-k filter (HF drops the warper at 0) */
  unsigned long long seed;   /* Philox key when exp_noise == NULL */
  int row_offset;            /* global index of candidate 0 of this rank */
  const float* exp_noise;    /* optional f32 [max_new][B][vocab] Exp(1) draws: multinomial == argmax(p/q) */
  const unsigned long long* group_seeds;  /* optional HOST array [n_groups]: Philox key per group (NULL: `seed` for every group) */
  float typical_mass;        /* tts(typical_sampling=True, typical_mass=.9) (api.py:361-364): 0 < typical_mass < 1 runs the reference's
                              * TypicalLogitsWarper (tortoise/utils/typical_sampling.py:11-33) where generate() runs it - after the
                              * repetition penalty, before temperature / top-k / top-p (autoregressive.py:558); 0 = off (the default) */
} tt_sampling;

/* Replaces `self.inference_model.generate(... do_sample=True ...)` (autoregressive.py:560-563;
 * loop spec stream_generator.py:916-1000): samples B candidates for up to max_new tokens, stops
 * early when every row has emitted stop_mel_token.  codes int32 [B][max_new], pre-filled with
 * stop_mel_token past each row's end (api.py:425-426 padding).  The per-token step is replayed
 * from a hipGraph that stays on the handle: a later call with the same B, prefix length(s), sampling
 * scalars and exp_noise replays it as it is - seeds, row_offset and the caller's `codes` buffer are
 * data (the sampler fills a buffer the handle owns; the finished columns are copied out at the end) -
 * anything else re-captures.  Synchronises `stream` before returning; *n_steps_host = tokens per row. */
int tt_ar_generate(tt_ar* h, int B, int max_new, const tt_sampling* s, int* codes, int* n_steps_host, void* stream);
/* The same loop in resumable pieces (the streaming path, api_fast.py:389-420 pulls tokens from
 * get_generator() chunk by chunk): first != 0 starts a generation (token 0 from the prefill logits), first == 0 resumes it;
 * n_more further tokens are sampled into codes int32 [B][ldcodes] (the SAME buffer on every call).
 * *n_total_host = tokens per row so far, *finished_host = 1 once every row has emitted stop_mel_token. */
int tt_ar_generate_chunk(tt_ar* h, int B, int first, int n_more, int ldcodes, const tt_sampling* s, int* codes, int* n_total_host,
                         int* finished_host, void* stream);
/* The latent half of the (token, latent) pairs the reference's streaming generator yields (stream_generator.py:916-1000,
 * consumed at api_fast.py:402-411): out f32 [B][n][D], latent i = final_norm(ln_f(hidden state that produced the logits of
 * token i)) - latent 0 from the prefill's start-token row, latent i >= 1 from the decode step that fed token i - 1.  They are
 * filed by the decode steps themselves (no extra pass); handles with max_batch <= 8 only; n <= tokens generated so far. */
int tt_ar_stream_latents(tt_ar* h, int B, int n, float* out, void* stream);

/* Engine options of a handle (not part of the reference's surface; defaults in brackets):
 *   TT_AR_OPT_LOOKAHEAD  [6]  decode steps the host may launch ahead of the device (the loop is paced by progress words the
 *                             last kernel of a step publishes to pinned memory; no queue drain inside the loop) */
#define TT_AR_OPT_LOOKAHEAD 4
int tt_ar_set_option(tt_ar* h, int option, int value);
/* Operand-overflow guard: the row norms and the sampler count launches that met a non-finite value (an fp16 operand beyond 65504
 * upstream).  Returns the count as of the last finished tt_ar_generate[_chunk] / tt_ar_latents (>= 0; tt_last_error() then names
 * the stage) or a negative error; reset != 0 clears it.  The reference autocasts this stage to fp16 only under half=True
 * (api.py:413-414); the host side re-runs a tripped stage with bf16 operands. */
int tt_ar_guard(tt_ar* h, int reset);
/* Counters for tests: which = 0 decode-step graph captures so far, 1 queue drains the launch loop fell back to (expected 0),
 * 2 kernel launches of one decode step. */
int tt_ar_stat(tt_ar* h, int which);

/* Teacher-forced single steps for parity tests: tt_ar_begin resets per-sequence state for B rows
 * after a prefill; tt_ar_decode_step feeds tokens int32 [B] (KV-cached position rule of
 * autoregressive.py:145-149: mel position row = index + 1 for index >= 1) and leaves the logits for
 * tt_ar_get_logits.  At most max_new_tokens - 1 steps per tt_ar_begin (one KV slot each); more is an error. */
int tt_ar_begin(tt_ar* h, int B, void* stream);
int tt_ar_decode_step(tt_ar* h, const int* tokens, void* stream);

/* Replaces UnifiedVoice.forward(..., return_latent=True, clip_inputs=False) (autoregressive.py:454-506,
 * get_logits 417-431; called from api.py:521-524).  emb f32 [k][n][D] is the concatenated
 * [cond | text | mel] embedding; out f32 [k][n][D] = final_norm(ln_f(trunk(emb))) for every row
 * (the host slices the mel rows). */
int tt_ar_latents(tt_ar* h, const float* emb, int k, int n, float* out, void* stream);

/* ============================================================================================
 * CLVP scoring   (reference: tortoise/models/clvp.py:99-135, called from api.py:463)
 * ============================================================================================ */
typedef struct tt_clvp_layer {   /* one attention + one feed-forward sublayer */
  const float* attn_norm_g;
  const void* w_qkv;             /* T [3D][D] = [to_q; to_k; to_v] (no bias) */
  const void* w_out; const float* b_out;
  const float* ff_norm_g;
  const void* w_ff1; const float* b_ff1;   /* T [2*inner][D]  GEGLU proj, value / gate rows INTERLEAVED in strips of 16:
                                            * [value 0..15 | gate 0..15 | value 16..31 | ...] (pack.py geglu_interleave) */
  const void* w_ff2; const float* b_ff2;   /* T [D][inner] */
} tt_clvp_layer;
typedef struct tt_clvp_tower {
  const tt_clvp_layer* layers_host;
  const float* emb;              /* f32 [tokens][D] */
  const float* inv_freq;         /* f32 [rot/2] */
  const float* norm_g; const float* norm_b;  /* final LayerNorm */
  const void* w_latent;          /* T [latent][D] (no bias) */
} tt_clvp_tower;
typedef struct tt_clvp_config {
  int dtype, dim, latent_dim, depth, heads, ff_inner, rot_dim;
  int max_rows;                  /* max B * n tokens in one tower call */
} tt_clvp_config;
typedef struct tt_clvp tt_clvp;
int tt_clvp_create(const tt_clvp_config* cfg, const tt_clvp_tower* text, const tt_clvp_tower* speech,
                   const float* temperature, tt_clvp** out);
void tt_clvp_destroy(tt_clvp* h);
/* text int32 [T] (one prompt, evaluated once instead of B times — api.py:463 repeats it),
 * codes int32 [B][n] -> scores f32 [B]. */
int tt_clvp_score(tt_clvp* h, const int* text, int T, const int* codes, int B, int n, float* scores, void* stream);
/* G <= 16 utterances in ONE speech-tower pass (long-form reading, tortoise/read.py:66-71 ranks its chunks one call after the other):
 * texts int32 = the G token sequences back to back, T_host[g] their lengths (HOST array); codes int32 [G * N][n], candidates
 * [g * N, (g + 1) * N) belong to utterance g; scores f32 [G * N].  Every score equals tt_clvp_score's on that utterance alone, bit for bit. */
int tt_clvp_score_groups(tt_clvp* h, const int* texts, const int* T_host, int G, const int* codes, int N, int n, float* scores, void* stream);
int tt_clvp_guard(tt_clvp* h, int reset);  /* operand-overflow guard of this stage, see tt_ar_guard */

/* ============================================================================================
 * Stage 2 — DiffusionTts + SpacedDiffusion.p_sample_loop
 * (reference: tortoise/models/diffusion_decoder.py:232-322, tortoise/utils/diffusion.py:312-621,
 *  called from api.py:117-130 do_spectrogram_diffusion)
 * ============================================================================================ */
typedef struct tt_attn_block {       /* arch_util.AttentionBlock */
  const float* norm_g; const float* norm_b;
  const void* w_qkv; const float* b_qkv;    /* T [3C][C], rows reordered to [q|k|v][head][64] */
  const void* w_proj; const float* b_proj;  /* T [C][C] */
  const float* relpos;                      /* f32 [heads][129] = bias[bucket(clamp(k-q,-64,64))] * 8, or NULL */
} tt_attn_block;
typedef struct tt_res_block {        /* diffusion_decoder.ResBlock */
  const float* gn1_g; const float* gn1_b;
  const void* w_in; const float* b_in;      /* T [C][C]      in_layers.2 (1x1) */
  const float* gn2_g; const float* gn2_b;
  const void* w_out; const float* b_out;    /* T [C][3][C]   out_layers.3 (k3) */
} tt_res_block;
typedef struct tt_diff_config {
  int dtype, channels, heads, num_layers;   /* 1024, 16, 10 */
  int in_channels, in_pad;                  /* 100, 128 */
  int out_channels;                         /* 200 */
  int latent_channels;                      /* 1024 */
  int max_seq;                              /* max S */
  int max_codes;                            /* max M */
  int max_steps;
  int max_batch;                            /* utterances one tt_diff_sample_batch run may hold, 1 .. 16 (0 is read as 1) */
} tt_diff_config;
typedef struct tt_diff_weights {
  /* timestep-independent conditioning (diffusion_decoder.py:232-255) */
  const void* w_latent_conv; const float* b_latent_conv;   /* T [C][3][latent] latent_conditioner.0 */
  const tt_attn_block* latent_attn_host;                   /* 4 blocks */
  const float* code_norm_g; const float* code_norm_b;
  const float* uncond_emb;                                 /* f32 [C] */
  /* per-step network */
  const void* w_time1; const float* b_time1;               /* T [C][C] time_embed.0 */
  const void* w_time2; const float* b_time2;               /* T [C][C] time_embed.2 */
  const void* w_emb_all; const float* b_emb_all;           /* T [(3+L+3)*2C][C] all ResBlock emb_layers.1 stacked */
  const tt_res_block* res_host;                            /* 3 integrator + L layers + 3 tail */
  const tt_attn_block* attn_host;                          /* 3 integrator + L layers */
  const void* w_inp; const float* b_inp;                   /* T [C][3][in_pad] */
  const void* w_integ; const float* b_integ;               /* T [C][2C] */
  const float* out_gn_g; const float* out_gn_b;
  const void* w_final; const float* b_final;               /* T [out][3][C] */
} tt_diff_weights;
typedef struct tt_diff tt_diff;
int tt_diff_create(const tt_diff_config* cfg, const tt_diff_weights* w, tt_diff** out);
void tt_diff_destroy(tt_diff* h);

/* DiffusionTts.timestep_independent (diffusion_decoder.py:232-260), latent branch, eval mode.
 * latents f32 [M][latent]; cond f32 [2C] (scale | shift); interp_idx int32 [S] = nearest-neighbour
 * source row of F.interpolate.  Keeps the [S][C] conditioning inside the handle. */
int tt_diff_condition(tt_diff* h, const float* latents, int M, const float* cond, const int* interp_idx, int S, void* stream);
int tt_diff_get_code_emb(tt_diff* h, float* dst, void* stream);  /* f32 [S][C], tests */

typedef struct tt_diff_step {   /* float32 views of the float64 schedule tables (utils/diffusion.py:1237-1250) */
  int timestep;                 /* timestep_map[i] fed to the network */
  float min_log, max_log;       /* posterior_log_variance_clipped[i], log(betas[i]) */
  float cfk;                    /* conditioning_free_k * (1 - i / N)   (diffusion.py:378-384) */
  float sqrt_recip, sqrt_recipm1, coef1, coef2;
  float nonzero;                /* 0 for i == 0 */
} tt_diff_step;

/* One denoiser evaluation (DiffusionTts.forward, diffusion_decoder.py:262-322), conditioned and
 * unconditioned rows batched so weights stream once.  x f32 [S][in_channels] token-major;
 * out f32 [2][S][out_channels] (row 0 conditioned, row 1 conditioning_free).  Tests. */
int tt_diff_forward(tt_diff* h, const float* x, int timestep, int cond_free, float* out, void* stream);

/* SpacedDiffusion.p_sample_loop (utils/diffusion.py:533-621).  steps_host[n_steps] in the order they
 * run (i = N-1 ... 0).  x_T f32 [in_channels][S] and step_noise f32 [n_steps][in_channels][S]
 * use the reference's channels-first layout (step_noise[j] is the draw consumed by the j-th step run;
 * the entry for i == 0 is ignored).  mel_out f32 [in_channels][S] = denormalize_tacotron_mel(x_0)
 * (utils/audio.py:59-64).  Each step replays one hipGraph. */
int tt_diff_sample(tt_diff* h, const float* x_T, const float* step_noise, const tt_diff_step* steps_host, int n_steps,
                   int cond_free, float* mel_out, void* stream);

/* Several utterances through ONE denoiser pass per step (long-form reading: tortoise/read.py:66-71 renders its chunks one after the
 * other; BASELINE config #4 asks for them "batched through DiffusionTts").  The utterances may differ in length: every one is
 * laid out in a slot of S_pad >= max S_u positions and handled EXACTLY as if it ran alone - GroupNorm statistics cover its own S_u
 * positions, attention sees its own S_u keys, and the positions past S_u are kept at zero in every convolution operand, which is
 * precisely the zero padding its convolutions read at the sequence end (arch_util.py / diffusion_decoder.py Conv1d padding=1).
 *   tt_diff_batch_begin    : U utterances, common padded length S_pad
 *   tt_diff_condition_slot : tt_diff_condition for utterance u (its own M codes, its own S <= S_pad)
 *   tt_diff_sample_batch   : tt_diff_sample for all of them on one schedule; x_T[u] f32 [in][S_u], step_noise[u] f32
 *                            [n_steps][in][S_u], mel_out[u] f32 [in][S_u] (HOST arrays of U device pointers) */
int tt_diff_batch_begin(tt_diff* h, int U, int S_pad, void* stream);
int tt_diff_condition_slot(tt_diff* h, int u, const float* latents, int M, const float* cond, const int* interp_idx, int S, void* stream);
int tt_diff_sample_batch(tt_diff* h, int U, const float* const* x_T, const float* const* step_noise, const tt_diff_step* steps_host,
                         int n_steps, int cond_free, float* const* mel_out, void* stream);

/* Split sampling (the reference evaluates both rows on one device, utils/diffusion.py:340-384): the
 * conditioned and the conditioning-free denoiser rows of ONE utterance run on two GPUs.  Every participant holds the
 * full sampler state (x, schedule, noise) and evaluates one row per step; the host exchanges the rows (one all_gather
 * of f32 [S][out_channels] per step) and every participant applies the identical p_sample update, so the states stay
 * bit-identical without a second exchange.  Per step: split_forward -> exchange -> split_update.
 *   begin  : tt_diff_sample's setup for n_steps; row = 0 conditioned, 1 conditioning-free; captures the row's hipGraph
 *   forward: this participant's model row of the current step -> out_row f32 [S][out_channels]
 *   update : rows f32 [2][S][out_channels] (row 0 conditioned, row 1 conditioning-free); step_noise as tt_diff_sample;
 *            mel_out f32 [in_channels][S] is rewritten every step (the last write is x_0 denormalised)
 *   end    : waits for the handle's stream and frees the graph */
int tt_diff_split_begin(tt_diff* h, const float* x_T, const tt_diff_step* steps_host, int n_steps, int row, void* stream);
int tt_diff_split_forward(tt_diff* h, float* out_row, void* stream);
int tt_diff_split_update(tt_diff* h, const float* rows, const float* step_noise, float* mel_out, void* stream);
int tt_diff_split_end(tt_diff* h);
/* Operand-overflow guard of this stage (see tt_ar_guard): GroupNorm statistics / sampler inputs that came out non-finite, as of the
 * last finished sampling run (after the caller synchronised its stream).  The reference runs this stage in fp32 (api.py:540-560). */
int tt_diff_guard(tt_diff* h, int reset);
int tt_diff_stat(tt_diff* h, int which);  /* which = 0: sampler-step graph captures so far (the step graph stays on the handle) */
/* TT_DIFF_OPT_OVERLAP_PREPASS [1]: the conditioning_timestep_integrator of every step (diffusion_decoder.py:292-293; it depends on the
 * timestep and the conditioning, not on x_t) is evaluated in chunks of the schedule on a second stream WHILE the sampler loop walks
 * the steps whose chunks are complete; 0 = whole pre-pass first (one stream).  Same results either way. */
#define TT_DIFF_OPT_OVERLAP_PREPASS 1
/* TT_DIFF_OPT_FUSED_GN [1]: ResBlock in_layers (diffusion_decoder.py:60-80: GroupNorm32 -> SiLU -> 1x1 conv) as ONE launch - the conv's
 * GEMM normalises, activates and casts its own f32 A rows (csrc/gemm_gna.h); 0 = stand-alone apply launch + 16-bit tensor.  (The other
 * two GroupNorm sites were built fused as well and measured slower - profiles/r05_ab_fused_groupnorm.txt - and are not in the library.) */
#define TT_DIFF_OPT_FUSED_GN 2
int tt_diff_set_option(tt_diff* h, int option, int value);  /* option = TT_DIFF_OPT_*; switching TT_DIFF_OPT_FUSED_GN drops the kept step graph */

/* ============================================================================================
 * Stage 3 — UnivNetGenerator.inference   (reference: tortoise/models/vocoder.py:300-312, api.py:559)
 * ============================================================================================ */
typedef struct tt_voc_block {
  const float* w_convt; const float* b_convt;        /* f32 [32][32][2*stride] */
  const void* w_kp_in; const float* b_kp_in;         /* T [64][5][mel_pad] */
  const void* w_kp_res[6]; const float* b_kp_res[6]; /* T [64][3][64] x (3 blocks x 2 convs) */
  const void* w_kp_kernel; const float* b_kp_kernel; /* T [24576][3][64] */
  const void* w_kp_bias; const float* b_kp_bias;     /* T [256][3][64] */
  const float* w_conv[4]; const float* b_conv[4];    /* f32 [32][32][3], dilations 1,3,9,27 */
  int stride;
} tt_voc_block;
typedef struct tt_voc_config { int dtype, max_frames, mel_channels, mel_pad; } tt_voc_config;
typedef struct tt_voc_weights {
  const float* w_pre; const float* b_pre;     /* f32 [32][64][7] */
  const tt_voc_block* blocks_host;            /* 3 */
  const float* w_post; const float* b_post;   /* f32 [1][32][7] */
} tt_voc_weights;
typedef struct tt_voc tt_voc;
int tt_voc_create(const tt_voc_config* cfg, const tt_voc_weights* w, tt_voc** out);
void tt_voc_destroy(tt_voc* h);
/* mel f32 [mel_channels][S] (channels-first, as the diffusion stage emits it); z f32 [64][S+10];
 * audio f32 [S*256].  Pads 10 frames of -11.5129 and drops the matching tail (vocoder.py:303-311). */
int tt_voc_run(tt_voc* h, const float* mel, int S, const float* z, float* audio, void* stream);
/* operand-overflow guard of this stage (see tt_ar_guard): workgroups of the location-variable convolutions that met a non-finite predicted
 * kernel value, as of the end of the last tt_voc_run once the caller has synchronised its stream; reset != 0 clears it */
int tt_voc_guard(tt_voc* h, int reset);

/* ============================================================================================
 * Diagnostics (bench.py's roofline leg, the graph-vs-eager tests, counter passes).  Never used on the product path.
 * tt_prof_enable: while on, every kernel launch is timed - single-kernel launchers through the start / end timestamps of the
 * dispatch itself (hipExtLaunchKernelGGL; the clock a rocprofv3 kernel trace reads), launchers that enqueue several kernels by a
 * HIP-event bracket on their stream.  tt_prof_read: out[4] = {launches, total_ms, algorithmic_flops, algorithmic_bytes};
 * synchronises.  tt_graph_replay(0) makes the decode / sampler loops enqueue their kernels eagerly instead of capturing and
 * replaying a hipGraph (same kernels, same order; process-wide, default on); returns the previous setting.
 * ============================================================================================ */
int tt_graph_replay(int on);
int tt_prof_enable(int on);
int tt_prof_classes(void);
const char* tt_prof_class_name(int id);
int tt_prof_read(int id, double* out);

/* ============================================================================================
 * Conditioning front-end of the voice_samples path (replaces autoregressive.get_conditioning and
 * diffusion.get_conditioning, called from api.py:276 and api.py:289 get_conditioning_latents)
 * ============================================================================================ */
typedef struct tt_cond_config {
  int dtype;
  int ar_dim, ar_heads, ar_blocks;          /* 1024, 16, 6: ConditioningEncoder (autoregressive.py:204-228) */
  int ar_mel, ar_mel_pad;                   /* 80, 128 */
  int diff_channels, diff_heads, diff_blocks; /* 1024 (the embedder is 2x that wide), 16, 5: contextual_embedder (diffusion_decoder.py:186-192) */
  int diff_mel, diff_mel_pad;               /* 100, 128 */
  int max_frames;                           /* longest clip in mel frames */
} tt_cond_config;
typedef struct tt_cond_weights {
  const void* ar_w_init; const float* ar_b_init;   /* T [D][ar_mel_pad] conditioning_encoder.init (1x1) */
  const tt_attn_block* ar_attn_host;               /* ar_blocks blocks, 64-wide heads: QKV rows [q|k|v][head][64], relpos NULL */
  const void* diff_w_c0; const float* diff_b_c0;   /* T [C][3][diff_mel_pad]  contextual_embedder.0 (k3, stride 2) */
  const void* diff_w_c1; const float* diff_b_c1;   /* T [2C][3][C]            contextual_embedder.1 (k3, stride 2) */
  const tt_attn_block* diff_attn_host;             /* diff_blocks blocks over 2C channels; heads wider than 64 keep the reference's
                                                    * QKV row order [head][q|k|v][ch]; relpos f32 [heads][129] scaled by sqrt(ch) */
} tt_cond_weights;
typedef struct tt_cond tt_cond;
int tt_cond_create(const tt_cond_config* cfg, const tt_cond_weights* w, tt_cond** out);
void tt_cond_destroy(tt_cond* h);
/* ConditioningEncoder on ONE clip: mel f32 [ar_mel][T] (channels first, as api.py:271-276 builds it) -> out f32 [ar_dim] = h[:, :, 0];
 * UnifiedVoice.get_conditioning is the mean of these vectors over the clips. */
int tt_cond_ar_clip(tt_cond* h, const float* mel, int T, float* out, void* stream);
/* contextual_embedder on ONE clip: mel f32 [diff_mel][T] -> out_sum f32 [2C] = sum over the clip's *frames output frames
 * (DiffusionTts.get_conditioning = sum over all clips / total frames: the clips are concatenated along time and averaged). */
int tt_cond_diff_clip(tt_cond* h, const float* mel, int T, float* out_sum, int* frames, void* stream);

/* ============================================================================================
 * HiFi-GAN decoder of the streaming path (replaces hifi_decoder.inference(gpt_latents, auto_conditioning),
 * api_fast.py:420 tts_stream and api_fast.py:517 tts; hifigan_decoder.py:159-294)
 * ============================================================================================ */
#define TT_HIFI_MAX_STAGES 6
typedef struct tt_hifi_resblock {           /* hifigan_decoder.ResBlock1: convs1[d] (dilated) / convs2[d] per dilation d */
  const void* w1[3]; const float* b1[3];    /* T [Cp][k][Cp]  (Cp = channels padded to a multiple of 64, pad rows / columns zero) */
  const void* w2[3]; const float* b2[3];
} tt_hifi_resblock;
typedef struct tt_hifi_config {
  int dtype;
  int in_channels, cond_channels, initial_channel;   /* 1024, 1024, 512 */
  int num_stages; int up_factor[TT_HIFI_MAX_STAGES]; /* 4: 8, 8, 2, 2 (transposed-conv kernel = 2 * factor) */
  int num_kernels; int kernel_size[3];               /* 3: 3, 7, 11 */
  int num_dilations; int dilation[3];                /* 3: 1, 3, 5 */
  float lrelu_slope;                                 /* 0.1 */
  int max_latents;                                   /* longest latent sequence (mel codes) */
} tt_hifi_config;
typedef struct tt_hifi_weights {
  const void* w_pre; const float* b_pre;             /* T [C0][7][in]  conv_pre */
  const void* w_cond; const float* b_cond;           /* T [C0][cond]   cond_layer (1x1) */
  const void* w_up[TT_HIFI_MAX_STAGES];              /* T [u * Cp_out][2][Cp_in]: row r * Cp_out + co = (w[:, co, r + u] | w[:, co, r]) */
  const float* b_up[TT_HIFI_MAX_STAGES];             /* f32 [u * Cp_out] */
  const tt_hifi_resblock* res_host;                  /* num_stages * num_kernels blocks */
  const void* w_post; const float* b_post;           /* T [1][7][Cp_last], f32 [1]  conv_post */
} tt_hifi_weights;
typedef struct tt_hifi tt_hifi;
int tt_hifi_create(const tt_hifi_config* cfg, const tt_hifi_weights* w, tt_hifi** out);
void tt_hifi_destroy(tt_hifi* h);
/* frames after the two linear interpolations of HifiganGenerator.inference (x4, x24000/22050); samples = frames * prod(up_factor) */
int tt_hifi_output_frames(int n_latents);
/* latents f32 [T][in_channels] (GPT latents of ONE sequence), g f32 [cond_channels] (AR conditioning latent)
 * -> wav f32 [*n_samples] in [-1, 1] (tanh), *n_samples = tt_hifi_output_frames(T) * prod(up_factor) */
int tt_hifi_run(tt_hifi* h, const float* latents, int T, const float* g, float* wav, int* n_samples, void* stream);

/* ============================================================================================
 * CVVP scoring - the optional second ranking model of tts(cvvp_amount > 0)
 * (reference: tortoise/models/cvvp.py:63-131, built at api.py:252-257, called per conditioning clip at api.py:464-472:
 *  clip_results = cvvp * cvvp_amount + clvp * (1 - cvvp_amount); the CHANGELOG calls the model "removed", the call sites remain)
 * ============================================================================================ */
typedef struct tt_cvvp_tower {              /* cvvp.CollapsingTransformer (cvvp.py:19-51) */
  const tt_clvp_layer* layers_host;         /* HOST array [depth]: the x-transformers Encoder, same sublayer layout as CLVP's (ff_mult = 1: inner = dim) */
  const float* inv_freq;                    /* [rot_dim / 2] */
  const float* norm_g; const float* norm_b; /* ContinuousTransformerWrapper.norm (LayerNorm, xtransformers.py:1213) */
  const void* w_pre0; const float* b_pre0;  /* pre_combiner.0 (1x1 conv) T [dim][dim] */
  tt_attn_block attn;                       /* pre_combiner.1 AttentionBlock(dim, heads), relpos = NULL */
  const void* w_pre2; const float* b_pre2;  /* pre_combiner.2 (1x1 conv) T [dim][dim] */
  const void* w_latent;                     /* to_conditioning_latent / to_speech_latent T [dim][dim], no bias */
} tt_cvvp_tower;
typedef struct tt_cvvp_config {
  int dtype;
  int dim, heads, depth, rot_dim;           /* 512, 8, 8, 32 (latent_multiplier 1: latent width == dim) */
  int mel_channels, mel_pad;                /* 80, padded to a multiple of 64 for the first convolution's operand */
  int max_rows;                             /* candidates x codes of one call */
  int max_cond_frames;                      /* mel frames of one conditioning clip (api.py:73-84 pads / cuts clips to 132300 samples = 517 frames) */
} tt_cvvp_config;
typedef struct tt_cvvp_weights {
  tt_cvvp_tower cond, speech;               /* conditioning_transformer / speech_transformer */
  const void* w_cond0; const float* b_cond0;   /* cond_emb.0 Conv1d(mel, dim/2, k 5, stride 2, padding 2) as T [dim/2][5][mel_pad] */
  const void* w_cond1; const float* b_cond1;   /* cond_emb.1 Conv1d(dim/2, dim, k 3, stride 2, padding 1) as T [dim][3][dim/2] */
  const float* speech_emb;                  /* f32 [mel_codes][dim]  speech_emb.emb.weight */
  const float* temperature;                 /* f32 [1] */
} tt_cvvp_weights;
typedef struct tt_cvvp tt_cvvp;
int tt_cvvp_create(const tt_cvvp_config* cfg, const tt_cvvp_weights* w, tt_cvvp** out);
void tt_cvvp_destroy(tt_cvvp* h);
/* Replaces the loop `for cl in range(auto_conds.shape[1]): cvvp_accumulator += self.cvvp(auto_conds[:, cl].repeat(B, 1, 1), batch,
 * return_loss=False)` and the division by the clip count (api.py:464-468).  mels f32 [n_clips][mel_channels][T] (the voice's conditioning
 * clips, api.py:262-276; every clip T frames), codes int32 [B][n] (fix_autoregressive_output'ed candidates, values < mel_codes)
 * -> scores f32 [B].  A clip's conditioning latent is computed ONCE (the reference repeats the clip B times). */
int tt_cvvp_score(tt_cvvp* h, const float* mels, int n_clips, int T, const int* codes, int B, int n, float* scores, void* stream);
int tt_cvvp_guard(tt_cvvp* h, int reset);  /* operand-overflow guard of this stage, see tt_ar_guard */

#ifdef __cplusplus
}
#endif
#endif
