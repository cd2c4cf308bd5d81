// Stage 3: UnivNetGenerator.inference (reference: tortoise/models/vocoder.py:267-312).
// Mel-rate KernelPredictor convolutions run on MFMA (conv-GEMM); everything at the audio rate is
// f32 matrix-core / VALU work over channels-first rows (univnet.hip).  The predicted location-variable kernels are
// written ONCE, in the operand type, by the KernelPredictor GEMM's epilogue ([L][24576] per LVC block: 43 MB at 9.3 s of
// audio, f32 until round 5) and read once by the four LVC layers of the block, which widen them on the way into LDS.
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"

using namespace tt;

struct tt_voc {
  tt_voc_config cfg;
  tt_voc_weights w;
  std::vector<tt_voc_block> blocks;
  Arena arena;
  StreamBridge sb;
  float* c_cf = nullptr;     // [mel][L] padded mel, channels-first
  float* c_tm = nullptr;     // [L][mel] token-major
  void* c_t = nullptr;       // [L][mel_pad] T
  float* kp_h = nullptr;     // [L][64] f32 KernelPredictor hidden state
  void* kp_ht = nullptr;     // [L][64] T
  void* kp_t1 = nullptr;     // [L][64] T
  void* kernels = nullptr;   // [L][24576] T
  float* kbias = nullptr;    // [L][256]
  float* xa = nullptr;       // [32][T] ping
  float* xb = nullptr;       // [32][T] pong
  float* o = nullptr;        // [32][T] conv output
  int* guard = nullptr;      // [4] device counter: workgroups of the location-variable convolutions that met a non-finite predicted kernel value
  int* guard_host = nullptr; // pinned copy, refreshed at the end of every tt_voc_run
};

__global__ void voc_pad_mel_kernel(const float* mel, float* c, int S, int L, int C) {
  // vocoder.py:303-305: append 10 frames of -11.5129
  const int total = C * L;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < total; f += gridDim.x * blockDim.x) {
    const int ch = f / L, t = f % L;
    c[f] = t < S ? mel[(size_t)ch * S + t] : -11.5129f;
  }
}

extern "C" {

int tt_voc_create(const tt_voc_config* cfg, const tt_voc_weights* w, tt_voc** out) {
  TT_REQUIRE(cfg && w && out, "tt_voc_create: null argument");
  TT_REQUIRE(cfg->mel_pad % 64 == 0 && cfg->mel_pad >= cfg->mel_channels, "tt_voc_create: mel_pad must be a multiple of 64");
  tt_voc* e = new tt_voc();
  e->cfg = *cfg;
  e->w = *w;
  e->blocks.assign(w->blocks_host, w->blocks_host + 3);
  const size_t L = (size_t)cfg->max_frames + 10;
  size_t hop = 1;
  for (int i = 0; i < 3; ++i) hop *= e->blocks[i].stride;
  const size_t T = L * hop;
  int rc = e->sb.init();
  const size_t es = dtype_bytes(cfg->dtype);  // (4: the fp32 verification mode)
  if (!rc) rc = e->arena.alloc_t(&e->c_cf, L * cfg->mel_channels);
  if (!rc) rc = e->arena.alloc_t(&e->c_tm, L * cfg->mel_channels);
  if (!rc) rc = e->arena.alloc(&e->c_t, (L + 8) * cfg->mel_pad * es);
  if (!rc) rc = e->arena.alloc_t(&e->kp_h, (L + 8) * 64);
  if (!rc) rc = e->arena.alloc(&e->kp_ht, (L + 8) * 64 * es);
  if (!rc) rc = e->arena.alloc(&e->kp_t1, (L + 8) * 64 * es);
#if defined(TT_VOC_KERNELS_F32)  // A/B knob (build.py --variant): the round-5 form, predicted kernels materialised in f32
  if (!rc) rc = e->arena.alloc(&e->kernels, L * 24576 * 4 + 64);
#else
  if (!rc) rc = e->arena.alloc(&e->kernels, L * 24576 * es + 64);
#endif
  if (!rc) rc = e->arena.alloc_t(&e->kbias, L * 256);
  if (!rc) rc = e->arena.alloc_t(&e->xa, 32 * T);
  if (!rc) rc = e->arena.alloc_t(&e->xb, 32 * T);
  if (!rc) rc = e->arena.alloc_t(&e->o, 32 * T);
  if (!rc) rc = e->arena.alloc_t(&e->guard, 4);
  if (!rc && hipHostMalloc((void**)&e->guard_host, 4 * sizeof(int)) != hipSuccess) { set_error("tt_voc_create: hipHostMalloc failed"); rc = -2; }
  if (!rc) e->guard_host[0] = 0;
  if (rc) {
    tt_voc_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

void tt_voc_destroy(tt_voc* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  if (e->guard_host) (void)hipHostFree(e->guard_host);
  e->arena.release();
  e->sb.destroy();
  delete e;
}

// Operand-overflow guard of this stage (see tt_ar_guard): workgroups that met non-finite predicted kernels (an fp16 KernelPredictor operand
// beyond 65504 upstream - behind the sigmoid * tanh gate and the final tanh the waveform itself would come out finite), as of the end of the
// last tt_voc_run after the caller synchronised its stream.  reset != 0 clears it.
int tt_voc_guard(tt_voc* e, int reset) {
  if (!e) { set_error("tt_voc_guard: null handle"); return -1; }
  const int n = e->guard_host[0];
  if (n > 0) set_error("vocoder stage: %d workgroup(s) met non-finite predicted kernels (operand overflow in %s)", n, e->cfg.dtype == DT_F16 ? "fp16: re-run this stage with bf16 operands" : "bf16");
  if (reset && n > 0) {
    if (hipMemsetAsync(e->guard, 0, 4 * sizeof(int), e->sb.own) != hipSuccess || hipStreamSynchronize(e->sb.own) != hipSuccess) { set_error("tt_voc_guard: reset failed"); return -2; }
    e->guard_host[0] = 0;
  }
  return n;
}

int tt_voc_run(tt_voc* e, const float* mel, int S, const float* z, float* audio, void* stream) {
  TT_REQUIRE(e && mel && z && audio, "tt_voc_run: null argument");
  TT_REQUIRE(S >= 1 && S <= e->cfg.max_frames, "tt_voc_run: %d frames exceed capacity %d", S, e->cfg.max_frames);
  hipStream_t us = (hipStream_t)stream, s = e->sb.own;
  TT_TRY(e->sb.enter(us));
  const int dt = e->cfg.dtype, MC = e->cfg.mel_channels, MP = e->cfg.mel_pad;
  const int L = S + 10;
  voc_pad_mel_kernel<<<std::min(cdiv(MC * L, 256), 2048), 256, 0, s>>>(mel, e->c_cf, S, L, MC);
  TT_CHECK_HIP(hipGetLastError());
  TT_TRY(transpose_launch(e->c_cf, e->c_tm, MC, L, s));
  TT_TRY(cast_pad_launch(dt, e->c_tm, MC, e->c_t, MP, L, MC, MP, s));
  // conv_pre: Conv1d(64 -> 32, k7, reflect)   (vocoder.py:255-256, 273)
  Conv1dArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.x = z; ca.w = e->w.w_pre; ca.bias = e->w.b_pre; ca.y = e->xa; ca.Cin = 64; ca.Cout = 32; ca.T = L; ca.k = 7; ca.dilation = 1;
  ca.reflect = 1; ca.in_slope = -1.f; ca.out_act = ACT_NONE;
  TT_TRY(conv1d_direct_launch(ca, s));
  float* x = e->xa;
  float* xn = e->xb;
  int hop = 1, T = L;
  static const int dil[4] = {1, 3, 9, 27};
  for (int bi = 0; bi < 3; ++bi) {
    const tt_voc_block& b = e->blocks[bi];
    // convt_pre: LeakyReLU + ConvTranspose1d   (vocoder.py:128-132, 162)
    ConvT1dArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.x = x; ta.w = b.w_convt; ta.bias = b.b_convt; ta.y = xn; ta.C = 32; ta.Tin = T; ta.stride = b.stride; ta.in_slope = 0.2f;
    TT_TRY(convt1d_launch(ta, s));
    { float* t = x; x = xn; xn = t; }
    hop *= b.stride;
    T = L * hop;
    // KernelPredictor(c)   (vocoder.py:66-93)
    GemmArgs g = gemm_args(e->c_t, MP, b.w_kp_in, 5 * MP, L, 64, 5 * MP);
    g.taps = 5; g.seq_len = L; g.bias = b.b_kp_in; g.act = ACT_LRELU; g.slope = 0.2f;
    g.out_f32 = e->kp_h; g.ldo32 = 64; g.out_t = e->kp_ht; g.ldot = 64;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    for (int r = 0; r < 3; ++r) {
      g = gemm_args(e->kp_ht, 64, b.w_kp_res[2 * r], 192, L, 64, 192);
      g.taps = 3; g.seq_len = L; g.bias = b.b_kp_res[2 * r]; g.act = ACT_LRELU; g.slope = 0.2f; g.out_t = e->kp_t1; g.ldot = 64;
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
      g = gemm_args(e->kp_t1, 64, b.w_kp_res[2 * r + 1], 192, L, 64, 192);
      g.taps = 3; g.seq_len = L; g.bias = b.b_kp_res[2 * r + 1]; g.act = ACT_LRELU; g.slope = 0.2f;
      g.res = e->kp_h; g.ldres = 64; g.out_f32 = e->kp_h; g.ldo32 = 64; g.out_t = e->kp_ht; g.ldot = 64;
      TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    }
    g = gemm_args(e->kp_ht, 64, b.w_kp_kernel, 192, L, 24576, 192);
    g.taps = 3; g.seq_len = L; g.bias = b.b_kp_kernel;
#if defined(TT_VOC_KERNELS_F32)
    g.out_f32 = (float*)e->kernels; g.ldo32 = 24576;
#else
    g.out_t = e->kernels; g.ldot = 24576;
#endif
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    g = gemm_args(e->kp_ht, 64, b.w_kp_bias, 192, L, 256, 192);
    g.taps = 3; g.seq_len = L; g.bias = b.b_kp_bias; g.out_f32 = e->kbias; g.ldo32 = 256;
    TT_TRY(gemm_launch(dt, EPI_STD, g, s));
    for (int j = 0; j < 4; ++j) {
      // conv_blocks[j]: LeakyReLU + dilated Conv1d, then LeakyReLU   (vocoder.py:134-146, 172-173)
      memset(&ca, 0, sizeof(ca));
      ca.x = x; ca.w = b.w_conv[j]; ca.bias = b.b_conv[j]; ca.y = e->o; ca.Cin = 32; ca.Cout = 32; ca.T = T; ca.k = 3;
      ca.dilation = dil[j]; ca.reflect = 0; ca.in_slope = 0.2f; ca.out_act = ACT_LRELU; ca.out_slope = 0.2f;
      TT_TRY(conv1d_direct_launch(ca, s));
      LvcArgs la;
      memset(&la, 0, sizeof(la));
      la.x_in = e->o; la.kernels = e->kernels; la.dtype = dt; la.ldk = 24576; la.koff = j * 6144; la.bias = e->kbias; la.ldb = 256; la.boff = j * 64;
#if defined(TT_VOC_KERNELS_F32)
      la.dtype = DT_F32;
#endif
      la.x = x; la.L = L; la.hop = hop; la.in_slope = -1.f; la.guard = e->guard;
      TT_TRY(lvc_launch(la, s));
    }
  }
  // conv_post: LeakyReLU + Conv1d(32 -> 1, k7, reflect) + tanh   (vocoder.py:258-262); drop the 10 pad frames
  memset(&ca, 0, sizeof(ca));
  ca.x = x; ca.w = e->w.w_post; ca.bias = e->w.b_post; ca.y = e->o; ca.Cin = 32; ca.Cout = 1; ca.T = T; ca.k = 7; ca.dilation = 1;
  ca.reflect = 1; ca.in_slope = 0.2f; ca.out_act = 5;
  TT_TRY(conv1d_direct_launch(ca, s));
  TT_CHECK_HIP(hipMemcpyAsync(audio, e->o, (size_t)S * hop * sizeof(float), hipMemcpyDeviceToDevice, s));
  TT_CHECK_HIP(hipMemcpyAsync(e->guard_host, e->guard, sizeof(int), hipMemcpyDeviceToHost, s));
  return e->sb.leave(us);
}

}  // extern "C"
