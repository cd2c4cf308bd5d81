// Conditioning front-end of the voice_samples path: the two encoders that turn a voice clip's mel
// spectrograms into the conditioning latents every later stage consumes (reference: tortoise/api.py:258-299).
//   * UnifiedVoice.get_conditioning   = ConditioningEncoder (tortoise/models/autoregressive.py:204-228, 444-452):
//       conv1x1(80 -> D) -> 6 AttentionBlocks (16 heads x 64, no relative positions) -> time step 0; mean over the clips.
//   * DiffusionTts.get_conditioning   = contextual_embedder (tortoise/models/diffusion_decoder.py:186-192, 222-230):
//       conv k3 stride 2 (100 -> C) -> conv k3 stride 2 (C -> 2C) -> 5 AttentionBlocks (16 heads x 128, T5 relative
//       positions); the clips are concatenated along time and averaged.
// Both run once per voice (the reference caches the result per voice as a .pth file, utils/audio.py:104-124), so nothing
// here is tuned: the stages reuse the engine's GEMM / GroupNorm / flash kernels through the same token-major layout as
// the denoiser, a stride-2 convolution is the stride-1 tap GEMM followed by an even-row gather, and the 128-wide heads of
// the embedder (the flash kernels are built for 64) use a small wave-per-query kernel.  One clip per call: the host
// averages the per-clip results exactly as the reference does (mean of the clip vectors / sum over frames of all clips
// divided by the total frame count).
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"

using namespace tt;

namespace {

// Generic small attention, QKVAttentionLegacy semantics (arch_util.py:44-77): one wave per (query, head).
//   qkv T [n][ldq]: head h keeps q at column h*3*ch, k at +ch, v at +2*ch (the reference's own channel order).
//   out T [n][ldo]: head h at columns h*ch ..  (== the [B, C, S] reshape of arch_util.py:75).
//   relpos f32 [heads][129] additive bias indexed by clamp(key - query, -64, 64) + 64, or null.
template <typename T>
__global__ __launch_bounds__(64) void attn_small_kernel(const T* __restrict__ qkv, int ldq, T* __restrict__ out, int ldo, int n, int ch,
                                                        const float* __restrict__ relpos, float scale) {
  extern __shared__ float sc[];  // [n] scores, then [ch] query
  float* qs = sc + n;
  const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const T* q = qkv + (size_t)i * ldq + (size_t)h * 3 * ch;
  for (int d = lane; d < ch; d += 64) qs[d] = (float)q[d];
  __syncthreads();
  float mx = -1e30f;
  for (int j = lane; j < n; j += 64) {
    const T* k = qkv + (size_t)j * ldq + (size_t)h * 3 * ch + ch;
    float s = 0.f;
    for (int d = 0; d < ch; ++d) s += qs[d] * (float)k[d];
    s *= scale;
    if (relpos) {
      int dd = j - i;
      dd = dd < -64 ? -64 : (dd > 64 ? 64 : dd);
      s += relpos[h * 129 + dd + 64];
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = 1.0f / sum;
  for (int d = lane; d < ch; d += 64) {
    float o = 0.f;
    for (int j = 0; j < n; ++j) o += sc[j] * (float)qkv[(size_t)j * ldq + (size_t)h * 3 * ch + 2 * ch + d];
    out[(size_t)i * ldo + (size_t)h * ch + d] = (T)(o * inv);
  }
}

__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += x[(size_t)r * C + c];
  out[c] = s;
}

__global__ void even_rows_index_kernel(int* idx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = 2 * i;
}

}  // namespace

struct tt_cond {
  tt_cond_config cfg;
  tt_cond_weights w;
  std::vector<tt_attn_block> ar_attn, diff_attn;
  Arena arena;
  StreamBridge sb;
  int rows = 0, cmax = 0;
  float* mel_t = nullptr;   // [T][mel] token-major f32
  void* mel_op = nullptr;   // [T][mel_pad] T
  float* ha = nullptr;      // [rows][cmax] f32 stream
  float* hb = nullptr;
  void* act = nullptr;      // [rows][cmax] T
  void* act2 = nullptr;     // [rows][cmax] T
  void* qkv = nullptr;      // [rows][3*cmax] T (generic attention)
  void* q = nullptr; void* k = nullptr; void* vt = nullptr;  // flash layouts (64-wide heads)
  void* att = nullptr;      // [rows][cmax] T
  float* gn_partial = nullptr;
  int* even_idx = nullptr;  // [rows] = 2 i
};

static int cond_gn(tt_cond* e, const float* x, int S, int C, const float* g, const float* b, void* out_t, hipStream_t s) {
  GroupNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.B = 1; a.S = S; a.C = C; a.gamma = g; a.beta = b; a.eps = 1e-5f; a.act = ACT_NONE;
  a.out_t = out_t; a.ldot = C; a.partial = e->gn_partial;
  return groupnorm_launch(e->cfg.dtype, a, s);
}

// AttentionBlock (arch_util.py:80-123): out = in + proj(attn(qkv(GN(in)))), token-major [S][C]
static int cond_attn_block(tt_cond* e, const tt_attn_block& w, const float* in, int S, int C, int heads, float* out, hipStream_t s) {
  const int dt = e->cfg.dtype, ch = C / heads;
  TT_TRY(cond_gn(e, in, S, C, w.norm_g, w.norm_b, e->act, s));
  if (ch == 64) {  // the engine's flash path: QKV rows packed [q|k|v][head][64], epilogue scatters into the attention layouts
    const int n_pad = round_up(S, 32);
    GemmArgs g = gemm_args(e->act, C, w.w_qkv, C, S, 3 * C, C);
    g.bias = w.b_qkv; g.seq_len = S; g.dmodel = C; g.heads = heads; g.q = e->q; g.k = e->k; g.vt = e->vt; g.seq_pad = n_pad;
    g.q_scale = 0.125f;  // (q * 64^-1/4) . (k * 64^-1/4)
    TT_TRY(gemm_launch(dt, EPI_QKV_HEADS, g, s));
    FlashArgs f;
    memset(&f, 0, sizeof(f));
    f.q = e->q; f.k = e->k; f.vt = e->vt; f.out = e->att; f.ldo = C; f.BH = heads; f.heads = heads; f.n = S; f.n_pad = n_pad;
    f.relpos = w.relpos;
    TT_TRY(flash_attention_launch(dt, f, s));
  } else {  // any other head width: QKV in the reference's own channel order, wave-per-query attention
    GemmArgs g = gemm_args(e->act, C, w.w_qkv, C, S, 3 * C, C);
    g.bias = w.b_qkv; g.out_t = e->qkv; g.ldot = 3 * C;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    const size_t smem = ((size_t)S + ch) * sizeof(float);
    TT_REQUIRE(smem <= 60 * 1024, "conditioning: %d frames x %d-wide heads exceed the attention score buffer", S, ch);
    const float scale = 1.0f / sqrtf((float)ch);
    if (dt == DT_BF16) attn_small_kernel<bf16><<<dim3(S, heads), 64, smem, s>>>((const bf16*)e->qkv, 3 * C, (bf16*)e->att, C, S, ch, w.relpos, scale);
    else attn_small_kernel<f16><<<dim3(S, heads), 64, smem, s>>>((const f16*)e->qkv, 3 * C, (f16*)e->att, C, S, ch, w.relpos, scale);
    TT_CHECK_HIP(hipGetLastError());
  }
  GemmArgs g = gemm_args(e->att, C, w.w_proj, C, S, C, C);
  g.bias = w.b_proj; g.res = in; g.ldres = C; g.out_f32 = out; g.ldo32 = C;
  return gemm_launch(dt, EPI_STD, g, s);
}

extern "C" {

int tt_cond_create(const tt_cond_config* cfg, const tt_cond_weights* w, tt_cond** out) {
  TT_REQUIRE(cfg && (cfg->dtype == DT_BF16 || cfg->dtype == DT_F16), "tt_cond_create: dtype must be TT_BF16 or TT_F16 (the fp32 verification mode covers the AR / CLVP / diffusion / vocoder stages)");
  TT_REQUIRE(cfg && w && out, "tt_cond_create: null argument");
  TT_REQUIRE(cfg->ar_dim % 64 == 0 && cfg->diff_channels % 64 == 0 && cfg->ar_mel_pad % 64 == 0 && cfg->diff_mel_pad % 64 == 0 &&
             cfg->ar_mel_pad >= cfg->ar_mel && cfg->diff_mel_pad >= cfg->diff_mel, "tt_cond_create: widths must be multiples of 64 (mel widths padded)");
  TT_REQUIRE(cfg->ar_dim % cfg->ar_heads == 0 && (2 * cfg->diff_channels) % cfg->diff_heads == 0 && cfg->max_frames >= 8, "tt_cond_create: bad shape");
  tt_cond* e = new tt_cond();
  e->cfg = *cfg;
  e->w = *w;
  e->ar_attn.assign(w->ar_attn_host, w->ar_attn_host + cfg->ar_blocks);
  e->diff_attn.assign(w->diff_attn_host, w->diff_attn_host + cfg->diff_blocks);
  e->rows = cfg->max_frames + 64;
  e->cmax = std::max(cfg->ar_dim, 2 * cfg->diff_channels);
  const size_t rows = e->rows, cm = e->cmax, melp = std::max(cfg->ar_mel_pad, cfg->diff_mel_pad);
  int rc = e->sb.init();
  if (!rc) rc = e->arena.alloc_t(&e->mel_t, rows * melp);
  if (!rc) rc = e->arena.alloc(&e->mel_op, rows * melp * 2);
  if (!rc) rc = e->arena.alloc_t(&e->ha, rows * cm);
  if (!rc) rc = e->arena.alloc_t(&e->hb, rows * cm);
  if (!rc) rc = e->arena.alloc(&e->act, rows * cm * 2);
  if (!rc) rc = e->arena.alloc(&e->act2, rows * cm * 2);
  if (!rc) rc = e->arena.alloc(&e->qkv, rows * 3 * cm * 2);
  if (!rc) rc = e->arena.alloc(&e->q, rows * cm * 2);
  if (!rc) rc = e->arena.alloc(&e->k, rows * cm * 2);
  if (!rc) rc = e->arena.alloc(&e->vt, (size_t)cm * (rows + 64) * 2);
  if (!rc) rc = e->arena.alloc(&e->att, rows * cm * 2);
  if (!rc) rc = e->arena.alloc_t(&e->gn_partial, groupnorm_partial_floats(1, (int)rows) + 64);
  if (!rc) rc = e->arena.alloc_t(&e->even_idx, rows);
  if (!rc) {
    even_rows_index_kernel<<<cdiv((int)rows, 256), 256>>>(e->even_idx, (int)rows);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      set_error("tt_cond_create: index fill failed");
      rc = -2;
    }
  }
  if (rc) {
    tt_cond_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_cond_destroy(tt_cond* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  e->arena.release();
  e->sb.destroy();
  delete e;
}

int tt_cond_ar_clip(tt_cond* e, const float* mel, int T, float* out, void* stream) {
  TT_REQUIRE(e && mel && out, "tt_cond_ar_clip: null argument");
  TT_REQUIRE(T >= 1 && T <= e->cfg.max_frames, "tt_cond_ar_clip: %d frames exceed capacity %d", T, e->cfg.max_frames);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int D = e->cfg.ar_dim, MC = e->cfg.ar_mel, MP = e->cfg.ar_mel_pad, dt = e->cfg.dtype;
  TT_TRY(transpose_launch(mel, e->mel_t, MC, T, s));                       // [mel][T] -> [T][mel]
  TT_TRY(cast_pad_launch(dt, e->mel_t, MC, e->mel_op, MP, T, MC, MP, s));
  GemmArgs g = gemm_args(e->mel_op, MP, e->w.ar_w_init, MP, T, D, MP);     // conditioning_encoder.init (1x1)
  g.bias = e->w.ar_b_init; g.out_f32 = e->ha; g.ldo32 = D;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  float* cur = e->ha;
  float* oth = e->hb;
  for (int i = 0; i < e->cfg.ar_blocks; ++i) {
    TT_TRY(cond_attn_block(e, e->ar_attn[i], cur, T, D, e->cfg.ar_heads, oth, s));
    float* t = cur; cur = oth; oth = t;
  }
  TT_CHECK_HIP(hipMemcpyAsync(out, cur, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, s));  // h[:, :, 0]
  return e->sb.leave(us);
}

int tt_cond_diff_clip(tt_cond* e, const float* mel, int T, float* out_sum, int* frames, void* stream) {
  TT_REQUIRE(e && mel && out_sum && frames, "tt_cond_diff_clip: null argument");
  TT_REQUIRE(e->cfg.diff_blocks > 0 && e->w.diff_w_c0 && e->w.diff_w_c1, "tt_cond_diff_clip: this handle was created without the diffusion embedder");
  TT_REQUIRE(T >= 4 && T <= e->cfg.max_frames, "tt_cond_diff_clip: %d frames outside [4, %d]", T, e->cfg.max_frames);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int C = e->cfg.diff_channels, C2 = 2 * C, MC = e->cfg.diff_mel, MP = e->cfg.diff_mel_pad, dt = e->cfg.dtype;
  const int T2 = (T + 1) / 2, T3 = (T2 + 1) / 2;  // Conv1d(k = 3, stride = 2, padding = 1): ceil(n / 2) outputs
  TT_TRY(transpose_launch(mel, e->mel_t, MC, T, s));
  TT_TRY(cast_pad_launch(dt, e->mel_t, MC, e->mel_op, MP, T, MC, MP, s));
  // stride-2 convolution = stride-1 tap GEMM at every position, then the even rows
  GemmArgs g = gemm_args(e->mel_op, MP, e->w.diff_w_c0, 3 * MP, T, C, 3 * MP);
  g.taps = 3; g.seq_len = T; g.bias = e->w.diff_b_c0; g.out_t = e->act; g.ldot = C;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  TT_TRY(gather_rows_launch((const float*)e->act, e->even_idx, (float*)e->act2, T2, C / 2, s));  // T rows of C elements == C / 2 words
  g = gemm_args(e->act2, C, e->w.diff_w_c1, 3 * C, T2, C2, 3 * C);
  g.taps = 3; g.seq_len = T2; g.bias = e->w.diff_b_c1; g.out_f32 = e->ha; g.ldo32 = C2;
  TT_TRY(gemm_launch(dt, EPI_STD, g, s));
  TT_TRY(gather_rows_launch(e->ha, e->even_idx, e->hb, T3, C2, s));
  float* cur = e->hb;
  float* oth = e->ha;
  for (int i = 0; i < e->cfg.diff_blocks; ++i) {
    TT_TRY(cond_attn_block(e, e->diff_attn[i], cur, T3, C2, e->cfg.diff_heads, oth, s));
    float* t = cur; cur = oth; oth = t;
  }
  colsum_kernel<<<cdiv(C2, 256), 256, 0, s>>>(cur, out_sum, T3, C2);
  TT_CHECK_HIP(hipGetLastError());
  *frames = T3;
  return e->sb.leave(us);
}

}  // extern "C"
