/* tortoise_mi355x_test.h - operator-level TEST entry points and A/B diagnostics of libtortoise_mi355x.so.
 *
 * NOT part of the drop-in boundary (include/tortoise_mi355x.h is what a maintainer binds): tests/ calls single kernels through these to
 * hold them against torch references, scripts/ uses the ttx_* switch for in-situ A/B runs.  Same conventions as the product header
 * (0 / negative return codes, device pointers, hipStream_t as void*).  Symbols here may change without an ABI version bump.
 */
#ifndef TORTOISE_MI355X_TEST_H
#define TORTOISE_MI355X_TEST_H
#include "tortoise_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide A/B switch of a kernel family; returns the previous value.  Set it before an engine captures its graphs.
 *   TTX_FLASH32   1 (default) = 32-query waves on v_mfma_f32_32x32x16 for non-causal sequences of more than 128 rows, 0 = 16-query waves
 *   TTX_GEMM_P8   1 (default) = the 8-wave eight-phase 256 x 256 tile (csrc/gemm_p8.h) where it applies, 0 = the 16-wave two-stage tile
 *                 (bit-identical results)
 *   TTX_VOC_MFMA  1 (default) = UnivNet's dilated 32 -> 32 convolutions and location-variable convolutions (hop 64 / 256) on
 *                 v_mfma_f32_32x32x2_f32 (exact f32), 0 = the thread-per-sample VALU kernels
 *   TTX_GEMM_SKINNY 1 (default) = 32 x 16 / 64 x 16 tiles for the weight-streaming GEMMs of decode batches of <= 64 rows, 0 = 64 x 64 tiles
 *                 (bit-identical results)
 *   TTX_AR_GEMV   a level: autoregressive handles created with max_batch <= 4 (the streaming engine) run the decode step's GEMMs
 *                 2 (default) = GEMV-shaped with the layer norms inside the QKV / c_fc launches (csrc/gemv.hip: five launches per layer),
 *                 1 = GEMV-shaped behind row-norm launches of their own (seven), 0 = on the MFMA tiles.  Read at tt_ar_create. */
#define TTX_FLASH32 0
#define TTX_GEMM_P8 1
#define TTX_VOC_MFMA 2
#define TTX_GEMM_SKINNY 3
#define TTX_AR_GEMV 4
int ttx_kernel_variant(int which, int v);

/* ============================================================================================
 * Operator-level entry points (used by tests/ to check single kernels against torch references)
 * ============================================================================================ */
int tt_op_gemm(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, int taps, int seq_len,
               int splitk, const float* bias, int act, const float* res, float* out_f32, void* out_t, void* stream);
int tt_op_layernorm(int dtype, const float* x, int M, int D, const float* g, const float* b, float eps, int rms,
                    void* out_t, float* out_f32, void* stream);
int tt_op_groupnorm(int dtype, const float* x, int B, int S, int C, const float* g, const float* b, const float* scale_shift,
                    int act, void* out_t, float* out_f32, float* workspace, void* stream);
size_t tt_op_groupnorm_workspace(int B, int S);
/* the fused ResBlock in_layers launch (TT_DIFF_OPT_FUSED_GN; diffusion_decoder.py:60-80): out_f32[B*S][N] = W . act(GroupNorm32(x)) + bias,
 * x f32 [B][S][1024] token-major, act = 3 (SiLU); 256 < B*S <= 4096, S >= 32, N % 256 == 0, 16-bit operand types */
int tt_op_gn_gemm(int dtype, const float* x, int B, int S, const float* gamma, const float* beta, int act, const void* W,
                  const float* bias, int N, float* out_f32, float* workspace, void* stream);
size_t tt_op_gn_gemm_workspace(int B, int S);
int tt_op_flash_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int heads, int n, int n_pad,
                          int causal, const float* relpos, void* stream);
/* the decode step's attention (HF GPT2Attention under tortoise/models/autoregressive.py:150-163, one query per (sequence, head)):
 * q T [B][heads * 64] pre-scaled by 1/8; shared prefix kp, vp T [heads][P1][64]; per-sequence caches kc T [B][heads][8][tmax][8]
 * (key-major 16-byte chunks), vc T [B][heads][tmax][64] with own keys 0 .. tgen - 1 valid; out T [B][heads * 64].
 * variant 0 = chosen from the shape, 1 = per-wave prefix kernel, 2 / 3 = shared-prefix kernel with 16 / 4 sequences per workgroup */
int tt_op_decode_attention(int dtype, const void* q, const void* kp, const void* vp, int P1, const void* kc, const void* vc, int tmax,
                           int tgen, void* out, int B, int heads, int variant, void* stream);
/* GEMV-shaped decode GEMM (csrc/gemv.hip): A T [M][K], W T [N][K], M <= 4, K in {1024, 2048, 4096}, N % 4 == 0;
 * epi 0: out_f32 [M][N] = A W^T + bias; 1: out_f32 += A W^T + bias (residual rows, in place); 2: out_t T [M][N] = gelu_tanh(A W^T + bias) */
int tt_op_gemv(int dtype, const void* A, const void* W, int M, int N, int K, const float* bias, int epi, float* out_f32, void* out_t, void* stream);
/* out_t[M][N] = gelu_tanh(LayerNorm(x[M][1024] f32; g, b, eps) W^T + bias): the same kernel with the layer norm inside */
int tt_op_gemv_ln(int dtype, const float* x, const float* g, const float* b, float eps, const void* W, int M, int N, const float* bias, void* out_t, void* stream);
int tt_op_sample(const float* logits, int ldl, int B, int V, unsigned* seen, const tt_sampling* s, int step, int* unfinished,
                 int stop_token, int* codes, int ldcodes, void* stream);
/* The typical-sampling mask alone (csrc/sampling.hip typical_mask_kernel; reference tortoise/utils/typical_sampling.py:11-33): out f32 [B][ldl]
 * = logits with every token outside the typical set of mass `mass` at -inf; the set is formed on the repetition-penalised scores (seen:
 * [B][(V+31)/32] bit mask of the ids generated so far), kept tokens carry their raw logit. */
int tt_op_typical_mask(const float* logits, int ldl, int B, int V, const unsigned* seen, float repetition_penalty, float mass, float* out,
                       void* stream);
int tt_op_conv1d(const float* x, const float* w, const float* bias, float* y, int Cin, int Cout, int T, int k, int dilation,
                 int reflect, float in_slope, int out_act, float out_slope, void* stream);
int tt_op_convt1d(const float* x, const float* w, const float* bias, float* y, int C, int Tin, int stride, float in_slope, void* stream);
/* location-variable convolution + gate (vocoder.py:182-216, 178-179); kernels [L][ldk] in the operand type `dtype` (TT_F32: float) */
int tt_op_lvc(int dtype, const float* x_in, const void* kernels, int ldk, int koff, const float* bias, int ldb, int boff, float* x, int L,
              int hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif
