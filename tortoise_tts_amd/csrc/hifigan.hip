// HiFi-GAN decoder of the streaming path: GPT latents -> waveform without CLVP / diffusion / UnivNet
// (reference: tortoise/models/hifigan_decoder.py:159-294 HifiganGenerator, called from tortoise/api_fast.py:420, 517).
//
// Unlike UnivNet (32 channels: VALU kernels), this generator is 512 -> 256 -> 128 -> 64 -> 32 channels wide with k = 3 / 7 / 11
// dilated convolutions: ~0.5 TFLOP per 9 s of audio, GEMM-shaped.  Everything therefore runs token-major ([samples][channels],
// channels padded to a multiple of 64 with zero weights) through the engine's MFMA conv-GEMM:
//   * a dilated conv1d is the tap GEMM with a row stride between taps (GemmArgs::dilation);
//   * ConvTranspose1d(k = 2u, stride u, padding u/2) is ONE 2-tap GEMM over the input rows with N = u * C_out: row j holds
//     the u output phases (t + u/2 = u j + r receives x[j] w[:, :, r] + x[j-1] w[:, :, r+u]), so the [rows+1][u * C] result
//     IS the upsampled [u * rows][C] tensor shifted by u/2 rows - no scatter, no zero insertion;
//   * every LeakyReLU rides an epilogue: conv1 of a ResBlock applies it to its output, conv2 / the transposed conv add the
//     skip and emit the activated operand copy of the next conv next to the f32 stream (GemmArgs::act_t);
//   * the multi-receptive-field mean of the three ResBlocks + the next LeakyReLU + the operand cast is one elementwise pass.
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"

using namespace tt;

namespace {

// F.interpolate(mode="linear", align_corners=False, scale_factor=s) along rows of a token-major tensor: rs = (float)(1 / s)
template <typename OT>
__global__ void interp_rows_kernel(const float* __restrict__ src, OT* __restrict__ dst, int Tin, int Tout, int C, float rs) {
  const int t = blockIdx.x;
  float pos = rs * ((float)t + 0.5f) - 0.5f;
  pos = pos < 0.f ? 0.f : pos;
  const int i0 = min((int)pos, Tin - 1), i1 = min(i0 + 1, Tin - 1);
  const float l1 = fminf(fmaxf(pos - (float)i0, 0.f), 1.f), l0 = 1.0f - l1;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    dst[(size_t)t * C + c] = (OT)(l0 * src[(size_t)i0 * C + c] + l1 * src[(size_t)i1 * C + c]);
}

// out[r][c] = T(lrelu((z0 + z1 + z2)[r][c] / nk, slope)), 4 channels per thread
template <typename T>
__global__ void mrf_combine_kernel(const float* __restrict__ z0, const float* __restrict__ z1, const float* __restrict__ z2, int nk,
                                   T* __restrict__ out, size_t n4, float slope) {
  const float inv = 1.0f / (float)nk;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = ((const float4*)z0)[i];
    if (nk > 1) { const float4 b = ((const float4*)z1)[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (nk > 2) { const float4 b = ((const float4*)z2)[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    float v[4] = {a.x * inv, a.y * inv, a.z * inv, a.w * inv};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * slope;
    ((typename Vec<T>::x4*)out)[i] = pack4<T>(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

struct tt_hifi {
  tt_hifi_config cfg;
  tt_hifi_weights w;
  std::vector<tt_hifi_resblock> res;
  Arena arena;
  StreamBridge sb;
  int cw[TT_HIFI_MAX_STAGES + 1];   // real channel width per level (level 0 = after conv_pre)
  int cp[TT_HIFI_MAX_STAGES + 1];   // padded to a multiple of 64
  float* lat1 = nullptr;   // [4 T][in] first interpolation
  void* lat2 = nullptr;    // [T2][in] T
  void* g_t = nullptr;     // [cond] T
  float* bias0 = nullptr;  // [c0] conv_pre bias + cond_layer(g)
  void* a_in = nullptr;    // [rows + 1][C] T: activated input of the next transposed conv (last row zero)
  float* o32 = nullptr;    // transposed-conv output, flat [(rows + 1) * u][C]
  void* o_t = nullptr;     // its activated operand copy
  float* xa = nullptr; float* xb = nullptr;
  float* z[3] = {nullptr, nullptr, nullptr};
  void* t1 = nullptr;      // lrelu(conv1) T
  void* xt = nullptr;      // lrelu(x) T between dilations
  size_t cap_elems = 0;    // element capacity of every stage buffer
};

static int hifi_elems(const tt_hifi* e, int T2, size_t* need) {
  size_t rows = T2, mx = (size_t)(T2 + 1) * e->cp[0];
  for (int i = 0; i < e->cfg.num_stages; ++i) {
    const size_t out_elems = (rows + 1) * e->cfg.up_factor[i] * e->cp[i + 1];
    mx = std::max(mx, out_elems);
    rows *= e->cfg.up_factor[i];
  }
  *need = mx;
  return 0;
}

extern "C" {

int tt_hifi_output_frames(int n_latents) {  // F.interpolate output lengths (hifigan_decoder.py:272-281): floor(n * 4), floor(. * 24000 / 22050)
  const int t1 = (int)floor((double)n_latents * (1024.0 / 256.0));
  return (int)floor((double)t1 * (24000.0 / 22050.0));
}

int tt_hifi_create(const tt_hifi_config* cfg, const tt_hifi_weights* w, tt_hifi** out) {
  TT_REQUIRE(cfg && (cfg->dtype == DT_BF16 || cfg->dtype == DT_F16), "tt_hifi_create: dtype must be TT_BF16 or TT_F16 (the fp32 verification mode covers the AR / CLVP / diffusion / vocoder stages)");
  TT_REQUIRE(cfg && w && out, "tt_hifi_create: null argument");
  TT_REQUIRE(cfg->num_stages >= 1 && cfg->num_stages <= TT_HIFI_MAX_STAGES && cfg->num_kernels >= 1 && cfg->num_kernels <= 3 &&
             cfg->num_dilations >= 1 && cfg->num_dilations <= 3, "tt_hifi_create: %d stages / %d kernels / %d dilations unsupported", cfg->num_stages, cfg->num_kernels, cfg->num_dilations);
  TT_REQUIRE(cfg->in_channels % 64 == 0 && cfg->cond_channels % 64 == 0 && cfg->initial_channel % 64 == 0 && cfg->max_latents >= 1, "tt_hifi_create: widths must be multiples of 64");
  tt_hifi* e = new tt_hifi();
  e->cfg = *cfg;
  e->w = *w;
  e->res.assign(w->res_host, w->res_host + cfg->num_stages * cfg->num_kernels);
  for (int i = 0; i <= cfg->num_stages; ++i) {
    e->cw[i] = cfg->initial_channel >> i;
    e->cp[i] = std::max(64, round_up(e->cw[i], 64));
  }
  for (int i = 0; i < cfg->num_stages; ++i)
    TT_REQUIRE(cfg->up_factor[i] >= 2 && cfg->up_factor[i] % 2 == 0 && e->cw[i + 1] >= 4, "tt_hifi_create: stage %d (factor %d, %d channels) unsupported", i, cfg->up_factor[i], e->cw[i + 1]);
  const int T2max = tt_hifi_output_frames(cfg->max_latents);
  size_t need = 0;
  hifi_elems(e, T2max, &need);
  e->cap_elems = need + 4096;
  int rc = e->sb.init();
  if (!rc) rc = e->arena.alloc_t(&e->lat1, (size_t)(4 * cfg->max_latents + 8) * cfg->in_channels);
  if (!rc) rc = e->arena.alloc(&e->lat2, (size_t)(T2max + 8) * cfg->in_channels * 2);
  if (!rc) rc = e->arena.alloc(&e->g_t, (size_t)cfg->cond_channels * 2 + 256);
  if (!rc) rc = e->arena.alloc_t(&e->bias0, cfg->initial_channel);
  if (!rc) rc = e->arena.alloc(&e->a_in, e->cap_elems * 2);
  if (!rc) rc = e->arena.alloc_t(&e->o32, e->cap_elems);
  if (!rc) rc = e->arena.alloc(&e->o_t, e->cap_elems * 2);
  if (!rc) rc = e->arena.alloc_t(&e->xa, e->cap_elems);
  if (!rc) rc = e->arena.alloc_t(&e->xb, e->cap_elems);
  for (int j = 0; j < 3 && !rc; ++j) rc = e->arena.alloc_t(&e->z[j], e->cap_elems);
  if (!rc) rc = e->arena.alloc(&e->t1, e->cap_elems * 2);
  if (!rc) rc = e->arena.alloc(&e->xt, e->cap_elems * 2);
  if (rc) {
    tt_hifi_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_hifi_destroy(tt_hifi* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  e->arena.release();
  e->sb.destroy();
  delete e;
}

int tt_hifi_run(tt_hifi* e, const float* latents, int T, const float* g, float* wav, int* n_samples, void* stream) {
  TT_REQUIRE(e && latents && g && wav && n_samples, "tt_hifi_run: null argument");
  TT_REQUIRE(T >= 1 && T <= e->cfg.max_latents, "tt_hifi_run: %d latents exceed capacity %d", T, e->cfg.max_latents);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const tt_hifi_config& c = e->cfg;
  const int dt = c.dtype, IN = c.in_channels, C0 = c.initial_channel;
  const int T1 = (int)floor((double)T * 4.0), T2 = tt_hifi_output_frames(T);
  TT_REQUIRE(T2 >= 1, "tt_hifi_run: no output frames");
  // latents -> x4 -> x24000/22050 (linear), operand type
  interp_rows_kernel<float><<<T1, 256, 0, s>>>(latents, e->lat1, T, T1, IN, (float)(1.0 / (1024.0 / 256.0)));
  if (dt == DT_BF16) interp_rows_kernel<bf16><<<T2, 256, 0, s>>>(e->lat1, (bf16*)e->lat2, T1, T2, IN, (float)(1.0 / (24000.0 / 22050.0)));
  else interp_rows_kernel<f16><<<T2, 256, 0, s>>>(e->lat1, (f16*)e->lat2, T1, T2, IN, (float)(1.0 / (24000.0 / 22050.0)));
  TT_CHECK_HIP(hipGetLastError());
  // conv_pre bias + cond_layer(g): one M = 1 GEMM, the conv_pre bias rides as the residual
  TT_TRY(cast_pad_launch(dt, g, c.cond_channels, e->g_t, c.cond_channels, 1, c.cond_channels, c.cond_channels, s));
  GemmArgs gm = gemm_args(e->g_t, c.cond_channels, e->w.w_cond, c.cond_channels, 1, C0, c.cond_channels);
  gm.bias = e->w.b_cond; gm.res = e->w.b_pre; gm.ldres = C0; gm.out_f32 = e->bias0; gm.ldo32 = C0;
  TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
  // conv_pre (k7) -> lrelu(0.1) -> operand of the first transposed conv ([T2 + 1][C0], last row zero)
  gm = gemm_args(e->lat2, IN, e->w.w_pre, 7 * IN, T2, C0, 7 * IN);
  gm.taps = 7; gm.seq_len = T2; gm.bias = e->bias0; gm.act = ACT_LRELU; gm.slope = c.lrelu_slope; gm.out_t = e->a_in; gm.ldot = e->cp[0];
  TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
  int rows = T2;
  for (int i = 0; i < c.num_stages; ++i) {
    const int u = c.up_factor[i], Cin = e->cp[i], C = e->cp[i + 1], p = u / 2;
    const int R = rows * u;  // samples after this stage
    TT_CHECK_HIP(hipMemsetAsync(offset_t(e->a_in, (size_t)rows * Cin), 0, (size_t)Cin * 2, s));  // x[rows] = 0
    // ConvTranspose1d as a 2-tap GEMM over rows + 1 input rows: flat output row t + p
    gm = gemm_args(e->a_in, Cin, e->w.w_up[i], 2 * Cin, rows + 1, u * C, 2 * Cin);
    gm.taps = 2; gm.seq_len = rows + 1; gm.bias = e->w.b_up[i]; gm.out_f32 = e->o32; gm.ldo32 = u * C; gm.out_t = e->o_t; gm.ldot = u * C;
    gm.act_t = ACT_LRELU; gm.slope_t = c.lrelu_slope;
    TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
    const float* o32 = e->o32 + (size_t)p * C;
    const void* o_t = offset_t(e->o_t, (size_t)p * C);
    for (int j = 0; j < c.num_kernels; ++j) {
      const tt_hifi_resblock& rb = e->res[i * c.num_kernels + j];
      const int ks = c.kernel_size[j];
      const float* x32 = o32;
      const void* xop = o_t;
      for (int d = 0; d < c.num_dilations; ++d) {
        const bool last = d == c.num_dilations - 1;
        gm = gemm_args(xop, C, rb.w1[d], ks * C, R, C, ks * C);   // convs1[d]: dilated, LeakyReLU on the output
        gm.taps = ks; gm.dilation = c.dilation[d]; gm.seq_len = R; gm.bias = rb.b1[d]; gm.act = ACT_LRELU; gm.slope = c.lrelu_slope;
        gm.out_t = e->t1; gm.ldot = C;
        TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
        float* xn = last ? e->z[j] : (d & 1 ? e->xb : e->xa);
        gm = gemm_args(e->t1, C, rb.w2[d], ks * C, R, C, ks * C);  // convs2[d] + skip; next dilation's activated operand
        gm.taps = ks; gm.seq_len = R; gm.bias = rb.b2[d]; gm.res = x32; gm.ldres = C; gm.out_f32 = xn; gm.ldo32 = C;
        if (!last) { gm.out_t = e->xt; gm.ldot = C; gm.act_t = ACT_LRELU; gm.slope_t = c.lrelu_slope; }
        TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
        x32 = xn;
        xop = e->xt;
      }
    }
    // mean of the ResBlocks -> LeakyReLU (0.1 between stages, 0.01 = F.leaky_relu default before conv_post) -> operand type
    const bool final_stage = i == c.num_stages - 1;
    const size_t n4 = (size_t)R * C / 4;
    const int blocks = (int)std::min<size_t>((n4 + 255) / 256, 8192);
    const float slope = final_stage ? 0.01f : c.lrelu_slope;
    if (dt == DT_BF16) mrf_combine_kernel<bf16><<<blocks, 256, 0, s>>>(e->z[0], e->z[1], e->z[2], c.num_kernels, (bf16*)e->a_in, n4, slope);
    else mrf_combine_kernel<f16><<<blocks, 256, 0, s>>>(e->z[0], e->z[1], e->z[2], c.num_kernels, (f16*)e->a_in, n4, slope);
    TT_CHECK_HIP(hipGetLastError());
    rows = R;
  }
  const int CL = e->cp[c.num_stages];
  gm = gemm_args(e->a_in, CL, e->w.w_post, 7 * CL, rows, 1, 7 * CL);  // conv_post (k7) -> tanh
  gm.taps = 7; gm.seq_len = rows; gm.bias = e->w.b_post; gm.act = ACT_TANH; gm.out_f32 = wav; gm.ldo32 = 1;
  TT_TRY(gemm_launch(dt, EPI_STD, gm, s));
  *n_samples = rows;
  return e->sb.leave(us);
}

}  // extern "C"
