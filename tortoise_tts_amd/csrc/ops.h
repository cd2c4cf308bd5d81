// Kernel launchers shared by the stage runners (gpt2 / clvp / diffusion / univnet) and by the
// operator-level C-ABI entry points the parity tests call.  All pointers are device pointers.
#pragma once
#include "common.h"
#include "gemm.h"

namespace tt {

// ------------------------------------------------------------------------------ row norms
enum NormMode { NORM_NONE = 0, NORM_LAYER = 1, NORM_RMS = 2 };
struct RowNormArgs {
  float* x;            // [M][ldx] f32 residual stream (read unless x_in; written when write_x)
  int ldx;
  const float* x_in;   // optional separate source rows
  int ldxin;
  int M, D;
  const float* add_bias;   // [D], added before the norm (bias of the preceding split-K GEMM)
  const float* add_slabs;  // [nslab][M][ldslab] partial sums of the preceding split-K GEMM
  int nslab;
  size_t slab_stride;
  int ldslab;
  int write_x;         // store the updated row back to x
  int mode;            // NormMode
  const float* g1;
  const float* b1;
  float eps1;
  const float* g2;     // optional second LayerNorm applied to the first one's output
  const float* b2;
  float eps2;
  void* out_t;         // normalised row in the GEMM operand type
  int ldot;
  float* out_f32;      // optional f32 copy of the normalised row
  int ldo32;
  // optional (generic kernel only): the f32 copy goes to out_f32 + (*f32_slot + f32_slot_base) * f32_slot_stride - a device-side
  // step counter selects the destination block, so a captured decode step can file its row under the step it belongs to
  const int* f32_slot;
  int f32_slot_base;
  size_t f32_slot_stride;
  int row_blocks;      // 1: always one workgroup per row (the decode step: a row's arithmetic order must not depend on how many rows the batch has)
  int* guard;          // optional device counter: += 1 per workgroup / wave that saw a non-finite input value (operand-overflow guard)
};
int rownorm_launch(int dtype, const RowNormArgs& a, hipStream_t stream);

// ------------------------------------------------------------------------------ group norm
struct GroupNormArgs {
  const float* x;  // [B][S][C]
  int B, S, C;
  const float* gamma;
  const float* beta;
  float eps;
  const float* scale_shift;  // optional [B?][2C]: y = y * (1 + scale) + shift
  size_t ss_batch_stride;    // 0 when every batch row shares one timestep embedding
  int ss_batch_div;          // batch row b reads block b / ss_batch_div (>= 1; 0 is treated as 1): rows of one timestep share it
  int act;
  void* out_t;
  int ldot;
  float* out_f32;
  int ldo32;
  float* partial;  // workspace, groupnorm_partial_floats(B, S) floats
  const float* gemm_part;  // optional: statistics already emitted by the producing GEMM's epilogue (gemm.hip
                           // run_epilogue) as [row_tile][2][C/16][2]; the separate statistics pass is skipped
  int part_rows;           // rows per row tile of gemm_part
  double inv_count;        // set by groupnorm_launch: 1 / (S * C / 32)
  // padded batches (several utterances of different lengths in one pass): sample b has vlen[b % vperiod] valid rows <= S; the
  // statistics cover the valid rows only and the rows beyond them are written as ZEROS (they are the zero padding the next
  // convolution sees past the end of a shorter sequence).  vperiod == 0: every sample has S rows.
  int vperiod;
  int vlen[32];
  int* guard;  // optional device counter: += 1 per (workgroup, group) whose statistics came out non-finite (operand-overflow guard)
};
int groupnorm_launch(int dtype, const GroupNormArgs& a, hipStream_t stream);
size_t groupnorm_partial_floats(int B, int S);

// ------------------------------------------------------------------------------ attention
struct FlashArgs {
  const void* q;   // [BH][n][64], pre-scaled by 1/sqrt(64)
  const void* k;   // [BH][n][64]
  const void* vt;  // [BH][64][n_pad]
  void* out;       // [B][n][ldo] with head h at columns h*64..
  int ldo;
  int BH, heads, n, n_pad;
  int causal;
  const float* relpos;  // optional [heads][129] additive bias indexed by clamp(key - query, -64, 64) + 64
  // padded batches: batch row b attends to its first nv[b % nv_period] keys only (and only those queries are computed); n stays
  // the row stride of the operands.  nv_period == 0: all n rows are valid.
  int nv_period;
  int nv[32];
  int variant;  // 0: chosen from the shape; 1: never the key-split form; 2: never the 32-query-wave kernel (microbenchmarks / A-B runs)
};
int flash_attention_launch(int dtype, const FlashArgs& a, hipStream_t stream);
int flash_f32_launch(const FlashArgs& a, hipStream_t stream);  // fp32 verification mode (attention_f32.hip)

struct DecodeAttnArgs {
  const void* q;    // [B][heads*64], pre-scaled
  const void* kp;   // shared prefix keys   [heads][P1][64]
  const void* vp;   // shared prefix values [heads][P1][64]
  int P1;
  const void* kc;   // per-sequence keys   [B][heads][8][tmax][8]
  const void* vc;   // per-sequence values [B][heads][tmax][64]
  int tmax;
  const int* step;  // device int: slot of the newest generated key (keys 0..*step are valid)
  int host_tgen;    // host-side copy of *step + 1 for profiling estimates only (0 when unknown)
  void* out;        // [B][heads*64]
  int B, heads;
  int variant;      // 0: chosen from the shape; 1: per-wave prefix kernel; 2 / 3: shared-prefix kernel with 16 / 4 sequences per workgroup
  // several utterances in one batch (ngroups > 1): sequence b belongs to group b / group_size, whose prefix K / V start
  // prefix_group_stride ELEMENTS after the previous group's and hold p1_tab[group] rows (P1 = the largest of them: LDS sizing)
  int ngroups, group_size;
  size_t prefix_group_stride;
  int p1_tab[16];
};
int decode_attention_launch(int dtype, const DecodeAttnArgs& a, hipStream_t stream);
int decode_attn_f32_launch(const DecodeAttnArgs& a, hipStream_t stream);  // fp32 verification mode (attention_f32.hip)

// ------------------------------------------------------------------------------ GEMV-shaped decode GEMMs (gemv.hip): M <= 4 rows
enum GemvEpi { GEMV_F32 = 0, GEMV_RES = 1, GEMV_GELU_T = 2, GEMV_QKV = 3 };
struct GemvArgs {
  const void* A;      // [M][lda] T activation rows
  int lda;
  const void* W;      // [N][ldw] T
  int ldw;
  int M, N, K;        // M <= 4; K in {1024, 2048, 4096}; N % 4 == 0
  const float* bias;  // [N] or null
  int epi;            // GemvEpi
  float* out_f32;     // GEMV_F32: [M][ldo32] = A W^T + bias; GEMV_RES: the residual rows, updated in place (x += A W^T + bias)
  int ldo32;
  void* out_t;        // GEMV_GELU_T: [M][ldot] T = gelu_tanh(A W^T + bias)
  int ldot;
  // GEMV_QKV (the decode step's QKV projection: q pre-scaled into qbuf, K / V appended at slot *step; layouts as DecodeAttnArgs)
  const int* step;
  void* qbuf;
  void* kc;
  void* vc;
  int heads, tmax, dmodel;
  float q_scale;
  // fused LayerNorm (GEMV_QKV / GEMV_GELU_T, K == 1024): when ln_x is set the activation rows are LayerNorm(ln_x[M][ldx] f32; ln_g, ln_b, ln_eps) and A is unused
  const float* ln_x;
  int ldx;
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  int* guard;  // counts rows whose variance is not finite (norm.hip's overflow guard), or null
};
bool gemv_supported(int dtype, const GemvArgs& a);
int gemv_launch(int dtype, const GemvArgs& a, hipStream_t stream);

// ------------------------------------------------------------------------------ AR sampling
struct SampleArgs {
  const float* logits;  // [B][ldl]; ldl == 0 broadcasts one row per utterance group (the shared-prefix prefill logits), ldg apart
  int ldl, ldg;
  int B, V;
  unsigned* seen;       // [B][(V+31)/32] bitmask of ids already in input_ids (repetition penalty)
  float rep_penalty, temperature, top_p;
  int top_k;
  const float* exp_noise;  // optional [max_steps][B][V] Exp(1) draws (parity runs); else Philox
  unsigned long long seed;
  int row_offset;          // global candidate index of row 0 (sharding-invariant Philox streams)
  const int* state;        // device: state[0] = tokens sampled so far
  int* unfinished;         // [B]
  int stop_token;
  int* codes;              // [B][ldcodes]
  int ldcodes;
  int* next_tok;           // [B]
  int* unfinished_count;   // [max_steps]: number of unfinished rows after each sampling step
  // optional fusion of the next decode step's embedding: embed_x[b][:] = tok_emb[tok][:] + pos_emb[step + pos_offset][:]
  float* embed_x;          // [B][D] or null
  const float* tok_emb;
  const float* pos_emb;
  int D, pos_offset;
  int pos_len;               // rows of pos_emb: the embedding of a token whose position lies beyond the table is skipped (it is never fed)
  // several utterances in one batch (ngroups > 1): row b is candidate b % group_size of group b / group_size; Philox key
  // group_seeds[group], broadcast logits (ldl == 0) row `group`
  int ngroups, group_size;
  unsigned long long group_seeds[16];
  // keys_dev != null: the Philox keys are read from device memory instead (keys_dev[group], keys_dev[0] without groups), so that a
  // captured step graph does not bake the seed of one call in and can be replayed by the next (tt_ar_generate)
  const unsigned long long* keys_dev;
  const int* row_offset_dev;  // != null: row_offset is read from device memory as well (same reason)
  int* guard;                 // optional device counter: += 1 per wave that read a NaN / +inf logit
  // typical sampling (tts(typical_sampling=True, typical_mass): tortoise/utils/typical_sampling.py, autoregressive.py:558): with
  // 0 < typical_mass < 1 a first launch writes the rows with every token outside the typical set at -inf to typical_out (laid out
  // like `logits`: same ldl / ldg), and the sampler reads those rows instead; 0 = off
  float typical_mass;
  float* typical_out;
};
int sample_launch(const SampleArgs& a, hipStream_t stream);
// the typical-sampling mask alone: rows of a.logits -> a.typical_out (sample_launch runs it ahead of the sampler when a.typical_mass != 0)
int typical_mask_launch(const SampleArgs& a, hipStream_t stream);
// state[0] += 1, state[1] = newest token's index; with `progress` (host-mapped int[2]) also publishes {tokens so far, first step
// after which unfinished_count was 0} for the host's launch loop (state[2] mirrors the latter on the device)
int ar_state_advance_launch(int* state, const int* unfinished_count, int* progress, hipStream_t stream);
int ar_begin_launch(int* state, unsigned* seen, int* unfinished, int* unfinished_count, int B, int V, int max_steps,
                    int start_token, hipStream_t stream);

// x[b][:] = tok_emb[tok[b]][:] + pos_emb[state[1] + pos_offset][:]   (pos_offset 2: kv_cache=True rule, 1: kv_cache=False rule)
int ar_embed_launch(const int* tok, const int* state, const float* tok_emb, const float* pos_emb, float* x, int B, int D, int pos_offset,
                    hipStream_t stream);

// ------------------------------------------------------------------------------ small fused ops
// (GEGLU is formed in the projection GEMM's epilogue: gemm.h EPI_GEGLU)
// x-transformers rotary on the first rot dims (rot = 32) of q, k ([BH][n][64]) and vt ([BH][64][n_pad])
int rotary_launch(int dtype, void* q, void* k, void* vt, const float* inv_freq, int BH, int n, int n_pad, int rot,
                  hipStream_t stream);
// dst[r][:] = src[idx[r]][:]   (f32 rows of C floats)
int gather_rows_launch(const float* src, const int* idx, float* dst, int rows, int C, hipStream_t stream);
// dst[b][:] = mean over n rows of src[b][n][:]
int mean_rows_launch(const float* src, float* dst, int B, int n, int C, hipStream_t stream);
// CLVP tail: out[b] = <normalize(t[b or 0]), normalize(s[b])> * exp(temperature)
int clvp_score_launch(const float* t, int t_rows, const float* s, const float* temperature, float* out, int B, int D,
                      hipStream_t stream);
// f32 -> T with optional column zero-padding: dst[r][0..cpad) = src[r][0..c) | 0
int cast_pad_launch(int dtype, const float* src, int lds, void* dst, int ldd, int rows, int c, int cpad, hipStream_t stream);
// dst[j][r][:] = src[r][:] for j < reps (f32 rows of C floats; r < rows)
int repeat_rows_launch(const float* src, float* dst, int rows, int reps, int C, hipStream_t stream);
// broadcast a [C] vector over rows
int broadcast_rows_launch(const float* vec, float* dst, int rows, int C, hipStream_t stream);
// f32 transpose: dst[c][r] = src[r][c]
int transpose_launch(const float* src, float* dst, int rows, int cols, hipStream_t stream);
// [rows][C] f32 -> SiLU -> T
int silu_cast_launch(int dtype, const float* src, void* dst, int n, hipStream_t stream);

// Diffusion p_sample epilogue (utils/diffusion.py:312-418, 487-531), token-major model output.
struct PSampleStep {    // same layout as tt_diff_step (include/tortoise_mi355x.h)
  int timestep;
  float min_log, max_log, cfk, sqrt_recip, sqrt_recipm1, coef1, coef2, nonzero;
};
struct PSampleArgs {
  const PSampleStep* steps;  // device array; entry *slot is used
  const int* slot;           // device int (advanced by slot_advance_launch)
  float* x;             // [S][C] f32 state, updated in place
  void* x_t;            // [2][S][cpad] operand copy for the next step's inp_block (both batch rows)
  int cpad;
  const float* out;     // [2][S][2C] model output rows: batch 0 = conditioned, batch 1 = unconditioned
  int has_uncond;
  const float* noise;   // [n_steps][C][S] channels-first draws (reference layout); entry *slot is used
  int S, C;
  float* mel_out;       // optional [C][S] channels-first denormalised mel written on the last step
  float mel_scale, mel_shift;
  int ld_rows;          // rows between the two batch rows of `out` / `x_t` (0: S; padded batches: the common padded length)
  // optional indirection (a sampler-step graph kept between calls must not bake caller pointers in): when `io` is set, noise and
  // mel_out are read from io[0] / io[1] of this utterance's device-side pointer pair instead of the two fields above
  const void* const* io;
  int* guard;           // optional device counter: += 1 per wave that read a non-finite model output (the x0 clamp would hide it)
};
int psample_launch(int dtype, const PSampleArgs& a, hipStream_t stream);
int slot_advance_launch(int* slot, const float* ss_all, float* ss_cur, int row_floats, int last_slot, hipStream_t stream);

// ------------------------------------------------------------------------------ UnivNet (fp32 VALU)
struct Conv1dArgs {
  const float* x;  // [Cin][T]
  const float* w;  // [Cout][Cin][k]
  const float* bias;
  float* y;        // [Cout][T]
  int Cin, Cout, T, k, dilation;
  int reflect;     // reflect padding (k/2 * dilation each side) instead of zeros
  float in_slope;  // LeakyReLU applied to the input when >= 0 (negative: none)
  int out_act;     // ACT_NONE / ACT_LRELU / 5 = tanh
  float out_slope;
};
int conv1d_direct_launch(const Conv1dArgs& a, hipStream_t stream);
struct ConvT1dArgs {
  const float* x;  // [C][Tin]
  const float* w;  // [Cin][Cout][2*stride]
  const float* bias;
  float* y;        // [C][Tin*stride]
  int C, Tin, stride;
  float in_slope;
};
int convt1d_launch(const ConvT1dArgs& a, hipStream_t stream);
struct LvcArgs {
  const float* x_in;    // [32][T] conv output (LeakyReLU applied here)
  const void* kernels;  // [L][ldk] row l holds layer j's [32][64][3] block at column koff, in the operand type `dtype` (round 6: the
  int ldk, koff;        // KernelPredictor GEMM writes its result once, in 16 bits; f32 in the fp32 verification mode / the operator tests)
  int dtype;            // DT_BF16 / DT_F16 / DT_F32
  const float* bias;    // [L][ldb], layer j's [64] block at column boff
  int ldb, boff;
  float* x;             // [32][T] residual stream: x += sigmoid(o[:32]) * tanh(o[32:])
  int L, hop;
  float in_slope;
  int* guard;           // optional device counter: += 1 per workgroup that staged a non-finite predicted kernel value (operand-overflow guard)
};
int lvc_launch(const LvcArgs& a, hipStream_t stream);

}  // namespace tt
