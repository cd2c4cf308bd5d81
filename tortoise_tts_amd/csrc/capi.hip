// Process-level and operator-level C-ABI entry points (include/tortoise_mi355x.h).
#include "runtime.h"
#include "../../include/tortoise_mi355x.h"
#include "../../include/tortoise_mi355x_test.h"

using namespace tt;
namespace tt { extern bool g_flash32; extern bool g_voc_mfma; extern bool g_gemm_p8; extern bool g_gemm_skinny; extern int g_ar_gemv; }  // attention.hip, univnet.hip, gemm.hip

extern "C" {

const char* tt_last_error(void) { return tt::last_error(); }
int tt_abi_version(void) { return 6; }  // ABI changes

int tt_init(void) {
  int dev = 0;
  TT_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TT_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  TT_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "tt_init: device is %s; this engine is built for gfx950 (MI355X) only", prop.gcnArchName);
  return gemm_init();
}

int tt_op_gemm(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, int taps, int seq_len, int splitk,
               const float* bias, int act, const float* res, float* out_f32, void* out_t, void* stream) {
  GemmArgs g = gemm_args(A, lda, W, ldw, M, N, K);
  g.taps = taps; g.seq_len = seq_len > 0 ? seq_len : M; g.splitk = splitk;
  g.bias = bias; g.act = act; g.slope = 0.2f; g.res = res; g.ldres = N; g.out_f32 = out_f32; g.ldo32 = N; g.out_t = out_t; g.ldot = N;
  return gemm_launch(dtype, EPI_STD, g, (hipStream_t)stream);
}

int tt_op_layernorm(int dtype, const float* x, int M, int D, const float* g, const float* b, float eps, int rms, void* out_t,
                    float* out_f32, void* stream) {
  RowNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (float*)x; a.ldx = D; a.M = M; a.D = D; a.mode = rms ? NORM_RMS : NORM_LAYER; a.g1 = g; a.b1 = b; a.eps1 = eps;
  a.out_t = out_t; a.ldot = D; a.out_f32 = out_f32; a.ldo32 = D;
  return rownorm_launch(dtype, a, (hipStream_t)stream);
}

size_t tt_op_groupnorm_workspace(int B, int S) { return groupnorm_partial_floats(B, S) * sizeof(float); }

int tt_op_groupnorm(int dtype, const float* x, int B, int S, int C, const float* g, const float* b, const float* scale_shift, int act,
                    void* out_t, float* out_f32, float* workspace, void* stream) {
  GroupNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.B = B; a.S = S; a.C = C; a.gamma = g; a.beta = b; a.eps = 1e-5f; a.scale_shift = scale_shift;
  a.ss_batch_stride = 2 * (size_t)C; a.act = act; a.out_t = out_t; a.ldot = C; a.out_f32 = out_f32; a.ldo32 = C; a.partial = workspace;
  return groupnorm_launch(dtype, a, (hipStream_t)stream);
}

// Test entry of the fused in_layers launch (gemm_gna.h): out_f32[B*S][N] = Linear(act(GroupNorm32(x)))(W, bias) for x f32 [B][S][1024].
// The statistics partials the kernel finalises are normally left by the producing GEMM's epilogue; here a helper kernel writes them in that
// layout (tiles of 32 rows, slot 1 = the rows of a tile that belong to the next sample, strips of 16 channels).
__global__ void gn_gemm_test_partials_kernel(const float* x, int M, int S, int C, float* part) {
  const int t = blockIdx.x, strip = threadIdx.x;  // one block per 32-row tile, one thread per 16-channel strip
  if (strip >= C / 16) return;
  const int b_first = (t * 32) / S, next_start = (b_first + 1) * S;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int r = t * 32; r < min(t * 32 + 32, M); ++r)
    for (int c = strip * 16; c < strip * 16 + 16; ++c) {
      const float v = x[(size_t)r * C + c];
      if (r < next_start) { s0 += v; q0 += v * v; } else { s1 += v; q1 += v * v; }
    }
  float* p = part + (((size_t)t * 2 + 0) * (C / 16) + strip) * 2;
  p[0] = s0; p[1] = q0;
  p = part + (((size_t)t * 2 + 1) * (C / 16) + strip) * 2;
  p[0] = s1; p[1] = q1;
}
size_t tt_op_gn_gemm_workspace(int B, int S) { return ((size_t)(B * S / 32 + 2) * 2 * 64 * 2 + 64) * 2 * sizeof(float); }

int tt_op_gn_gemm(int dtype, const float* x, int B, int S, const float* gamma, const float* beta, int act, const void* W, const float* bias, int N,
                  float* out_f32, float* workspace, void* stream) {
  TT_REQUIRE(x && gamma && beta && W && bias && out_f32 && workspace && B >= 1 && S >= 1, "tt_op_gn_gemm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int C = 1024, M = B * S;
  float* part_in = workspace;
  float* part_out = workspace + ((size_t)(M / 32 + 2) * 2 * 64 * 2 + 64);
  gn_gemm_test_partials_kernel<<<cdiv(M, 32), 64, 0, s>>>(x, M, S, C, part_in);
  TT_CHECK_HIP(hipGetLastError());
  GemmArgs g = gemm_args(x, C, W, C, M, N, C);
  g.bias = bias; g.out_f32 = out_f32; g.ldo32 = N; g.gn_part = part_out; g.gn_seq = S;
  GemmGnArgs n;
  memset(&n, 0, sizeof(n));
  n.gamma = gamma; n.beta = beta; n.gemm_part = part_in; n.part_rows = 32; n.S = S; n.eps = 1e-5f; n.act = act;
  TT_REQUIRE(gemm_gna_supported(dtype, EPI_STD, g, n), "tt_op_gn_gemm: no fused kernel for B=%d S=%d N=%d act=%d dtype=%d (256 < B*S <= 4096, S >= 32, N %% 256 == 0, SiLU, 16-bit operands)", B, S, N, act, dtype);
  return gemm_gna_launch(dtype, EPI_STD, g, n, s);
}

// Process-wide A/B switches of kernel families (include/tortoise_mi355x_test.h; like tt_graph_replay: set them before an engine captures
// its graphs - a kept graph replays the kernels it was captured with).  Returns the previous value.
int ttx_kernel_variant(int which, int v) {
  if (which == TTX_AR_GEMV) {  // a level, not a switch: 0 MFMA tiles | 1 GEMV launches behind the row-norm launches | 2 the layer norms inside the GEMVs
    TT_REQUIRE(v >= 0 && v <= 2, "ttx_kernel_variant(TTX_AR_GEMV): level %d", v);
    const int prev = tt::g_ar_gemv;
    tt::g_ar_gemv = v;
    return prev;
  }
  bool* sw = which == TTX_FLASH32 ? &tt::g_flash32 : which == TTX_GEMM_P8 ? &tt::g_gemm_p8 : which == TTX_VOC_MFMA ? &tt::g_voc_mfma : which == TTX_GEMM_SKINNY ? &tt::g_gemm_skinny : nullptr;
  TT_REQUIRE(sw != nullptr, "ttx_kernel_variant: unknown kernel family %d", which);
  const int prev = *sw ? 1 : 0;
  *sw = v != 0;
  return prev;
}

int tt_op_flash_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int heads, int n, int n_pad,
                          int causal, const float* relpos, void* stream) {
  FlashArgs f;
  memset(&f, 0, sizeof(f));
  f.q = q; f.k = k; f.vt = vt; f.out = out; f.ldo = heads * 64; f.BH = B * heads; f.heads = heads; f.n = n; f.n_pad = n_pad;
  f.causal = causal; f.relpos = relpos;
  return flash_attention_launch(dtype, f, (hipStream_t)stream);
}

// GEMV-shaped decode GEMM (gemv.hip; handles of <= 4 sequences): epi 0 = out_f32 = A W^T + bias, 1 = x (out_f32) += A W^T + bias, 2 = out_t = gelu_tanh(A W^T + bias)
int tt_op_gemv(int dtype, const void* A, const void* W, int M, int N, int K, const float* bias, int epi, float* out_f32, void* out_t, void* stream) {
  TT_REQUIRE(epi >= 0 && epi <= 2, "tt_op_gemv: epi %d (the QKV scatter is tested through the engine)", epi);
  GemvArgs v;
  memset(&v, 0, sizeof(v));
  v.A = A; v.lda = K; v.W = W; v.ldw = K; v.M = M; v.N = N; v.K = K; v.bias = bias; v.epi = epi; v.out_f32 = out_f32; v.ldo32 = N; v.out_t = out_t; v.ldot = N;
  return gemv_launch(dtype, v, (hipStream_t)stream);
}
// out_t[M][N] = gelu_tanh(LayerNorm(x[M][1024]; g, b, eps) W^T + bias): the GEMV with the layer norm inside (the decode step's c_fc at <= 4 sequences)
int tt_op_gemv_ln(int dtype, const float* x, const float* g, const float* b, float eps, const void* W, int M, int N, const float* bias, void* out_t, void* stream) {
  GemvArgs v;
  memset(&v, 0, sizeof(v));
  v.ln_x = x; v.ldx = 1024; v.ln_g = g; v.ln_b = b; v.ln_eps = eps; v.W = W; v.ldw = 1024; v.M = M; v.N = N; v.K = 1024; v.bias = bias; v.epi = GEMV_GELU_T; v.out_t = out_t; v.ldot = N;
  return gemv_launch(dtype, v, (hipStream_t)stream);
}

// The decode step's attention (HF GPT2Attention with a KV cache: one query per (sequence, head) over [shared prefix | own keys]) on
// caller-provided caches.  tgen own keys (slots 0 .. tgen - 1) are valid; the device-side step word the kernels read is made here.
int tt_op_decode_attention(int dtype, const void* q, const void* kp, const void* vp, int P1, const void* kc, const void* vc, int tmax, int tgen,
                           void* out, int B, int heads, int variant, void* stream) {
  TT_REQUIRE(q && kp && vp && kc && vc && out && tgen >= 1 && tgen <= tmax, "tt_op_decode_attention: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int* step = nullptr;
  TT_CHECK_HIP(hipMalloc((void**)&step, sizeof(int)));
  const int newest = tgen - 1;
  TT_CHECK_HIP(hipMemcpyAsync(step, &newest, sizeof(int), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipStreamSynchronize(s));
  DecodeAttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.kp = kp; a.vp = vp; a.P1 = P1; a.kc = kc; a.vc = vc; a.tmax = tmax; a.step = step; a.host_tgen = tgen;
  a.out = out; a.B = B; a.heads = heads; a.variant = variant;
  int rc = decode_attention_launch(dtype, a, s);
  hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(step);
  TT_TRY(rc);
  TT_CHECK_HIP(e);
  return 0;
}

// One sampling step on caller-provided state (seen bitmask, unfinished flags); `step` indexes codes / exp_noise.
int tt_op_sample(const float* logits, int ldl, int B, int V, unsigned* seen, const tt_sampling* sp, int step, int* unfinished,
                 int stop_token, int* codes, int ldcodes, void* stream) {
  TT_REQUIRE(sp, "tt_op_sample: null sampling parameters");
  hipStream_t s = (hipStream_t)stream;
  int* scratch = nullptr;  // state[2] | next_tok[B] | unfinished_count[step+1]
  const size_t n = 2 + (size_t)B + step + 1;
  TT_CHECK_HIP(hipMalloc((void**)&scratch, n * sizeof(int)));
  TT_CHECK_HIP(hipMemsetAsync(scratch, 0, n * sizeof(int), s));
  const int st[2] = {step, step - 1};
  TT_CHECK_HIP(hipMemcpyAsync(scratch, st, sizeof(st), hipMemcpyHostToDevice, s));
  TT_CHECK_HIP(hipStreamSynchronize(s));
  SampleArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.ldl = ldl; a.B = B; a.V = V; a.seen = seen;
  a.rep_penalty = sp->repetition_penalty; a.temperature = sp->temperature; a.top_p = sp->top_p; a.top_k = sp->top_k;
  a.exp_noise = sp->exp_noise; a.seed = sp->seed; a.row_offset = sp->row_offset;
  a.state = scratch; a.unfinished = unfinished; a.stop_token = stop_token; a.codes = codes; a.ldcodes = ldcodes;
  a.next_tok = scratch + 2; a.unfinished_count = scratch + 2 + B;
  float* typ = nullptr;
  if (sp->typical_mass != 0.f) {
    if (hipMalloc((void**)&typ, (size_t)B * (ldl ? ldl : V) * sizeof(float)) != hipSuccess) {
      (void)hipFree(scratch);
      set_error("tt_op_sample: hipMalloc failed");
      return -2;
    }
    a.typical_mass = sp->typical_mass; a.typical_out = typ;
  }
  int rc = sample_launch(a, s);
  hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(scratch);
  if (typ) (void)hipFree(typ);
  TT_TRY(rc);
  TT_CHECK_HIP(e);
  return 0;
}

int tt_op_typical_mask(const float* logits, int ldl, int B, int V, const unsigned* seen, float repetition_penalty, float mass, float* out,
                       void* stream) {
  SampleArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.ldl = ldl; a.B = B; a.V = V; a.seen = const_cast<unsigned*>(seen);
  a.rep_penalty = repetition_penalty; a.typical_mass = mass; a.typical_out = out;
  return typical_mask_launch(a, (hipStream_t)stream);
}

int tt_op_conv1d(const float* x, const float* w, const float* bias, float* y, int Cin, int Cout, int T, int k, int dilation, int reflect,
                 float in_slope, int out_act, float out_slope, void* stream) {
  Conv1dArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.Cin = Cin; a.Cout = Cout; a.T = T; a.k = k; a.dilation = dilation; a.reflect = reflect;
  a.in_slope = in_slope; a.out_act = out_act; a.out_slope = out_slope;
  return conv1d_direct_launch(a, (hipStream_t)stream);
}

int tt_op_convt1d(const float* x, const float* w, const float* bias, float* y, int C, int Tin, int stride, float in_slope, void* stream) {
  ConvT1dArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.C = C; a.Tin = Tin; a.stride = stride; a.in_slope = in_slope;
  return convt1d_launch(a, (hipStream_t)stream);
}

int tt_op_lvc(int dtype, const float* x_in, const void* kernels, int ldk, int koff, const float* bias, int ldb, int boff, float* x, int L, int hop,
              void* stream) {
  LvcArgs a;
  memset(&a, 0, sizeof(a));
  a.x_in = x_in; a.kernels = kernels; a.dtype = dtype; a.ldk = ldk; a.koff = koff; a.bias = bias; a.ldb = ldb; a.boff = boff; a.x = x; a.L = L; a.hop = hop;
  a.in_slope = -1.f;
  return lvc_launch(a, (hipStream_t)stream);
}

}  // extern "C"

// sizeof() of every struct that crosses the boundary, so host bindings can verify their mirrors
// without a GPU (tests/test_abi.py).
extern "C" size_t tt_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(tt_gpt_layer);
    case 1: return sizeof(tt_ar_config);
    case 2: return sizeof(tt_ar_weights);
    case 3: return sizeof(tt_sampling);
    case 4: return sizeof(tt_clvp_layer);
    case 5: return sizeof(tt_clvp_tower);
    case 6: return sizeof(tt_clvp_config);
    case 7: return sizeof(tt_attn_block);
    case 8: return sizeof(tt_res_block);
    case 9: return sizeof(tt_diff_config);
    case 10: return sizeof(tt_diff_weights);
    case 11: return sizeof(tt_diff_step);
    case 12: return sizeof(tt_voc_block);
    case 13: return sizeof(tt_voc_config);
    case 14: return sizeof(tt_voc_weights);
    case 15: return sizeof(tt_cond_config);
    case 16: return sizeof(tt_cond_weights);
    case 17: return sizeof(tt_hifi_resblock);
    case 18: return sizeof(tt_hifi_config);
    case 19: return sizeof(tt_hifi_weights);
    case 20: return sizeof(tt_cvvp_tower);
    case 21: return sizeof(tt_cvvp_config);
    case 22: return sizeof(tt_cvvp_weights);
  }
  return 0;
}
