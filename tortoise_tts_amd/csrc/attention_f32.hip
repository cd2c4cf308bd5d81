// fp32-operand attention of the VERIFICATION mode (DT_F32; tests only): the same contracts as flash_attention_launch /
// decode_attention_launch (ops.h) with q, k, v and the KV caches held in fp32 and plain VALU arithmetic - one wave per query row
// (full pass) or per (sequence, head) (decode step).  Slow by design; see gemm_f32.hip.
#include "ops.h"

namespace tt {

// q, k: [BH][n][64]; vt: [BH][64][n_pad]; out: [B][n][ldo] with head h at columns h*64.. ; one wave per query, 4 queries per block.
__global__ __launch_bounds__(256) void flash_f32_kernel(FlashArgs a) {
  __shared__ float pw[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh % a.heads;
  const int qi = blockIdx.x * 4 + wave;
  const int nvb = a.nv_period > 0 ? a.nv[b % a.nv_period] : a.n;  // keys (and queries) of this batch row
  if (qi >= nvb) return;
  const float* q = (const float*)a.q + ((size_t)bh * a.n + qi) * 64;
  const float* K = (const float*)a.k + (size_t)bh * a.n * 64;
  const float* VT = (const float*)a.vt + (size_t)bh * 64 * a.n_pad;
  float qr[64];
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const float4 t = *(const float4*)(q + d);
    qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
  }
  const int nk = a.causal ? min(qi + 1, nvb) : nvb;
  float m = -INFINITY, l = 0.f, o = 0.f;  // o: output dimension `lane`
  for (int k0 = 0; k0 < nk; k0 += 64) {
    const int key = k0 + lane;
    float s = -INFINITY;
    if (key < nk) {
      const float* kr = K + (size_t)key * 64;
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const float4 t = *(const float4*)(kr + d);
        acc = fmaf(qr[d], t.x, acc); acc = fmaf(qr[d + 1], t.y, acc); acc = fmaf(qr[d + 2], t.z, acc); acc = fmaf(qr[d + 3], t.w, acc);
      }
      if (a.relpos) acc += a.relpos[h * 129 + min(max(key - qi, -64), 64) + 64];
      s = acc;
    }
    const float mn = fmaxf(m, wave_max(s));
    const float p = key < nk ? expf(s - mn) : 0.f;
    const float scale = expf(m - mn);  // (m == -inf on the first chunk: exp(-inf) == 0)
    l = l * scale + wave_sum(p);
    o *= scale;
    pw[wave][lane] = p;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* vr = VT + (size_t)lane * a.n_pad + k0;
    const int cnt = min(64, nk - k0);
    for (int j = 0; j < cnt; ++j) o = fmaf(pw[wave][j], vr[j], o);
    __builtin_amdgcn_wave_barrier();
    m = mn;
  }
  ((float*)a.out)[((size_t)b * a.n + qi) * a.ldo + h * 64 + lane] = o / l;
}

int flash_f32_launch(const FlashArgs& a, hipStream_t stream) {
  ProfScope ps(PROF_FLASH, stream, 4.0 * a.BH * (double)a.n * a.n * 64.0, (double)a.BH * a.n * 64 * 4.0 * 4.0, true);
  launch_timed(ps, flash_f32_kernel, dim3(cdiv(a.n, 4), a.BH), dim3(256), 0, stream, a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// Decode step: one wave per (sequence, head); scores of the whole context in the LDS (ctx_cap floats per wave).
__global__ __launch_bounds__(256) void decode_attn_f32_kernel(DecodeAttnArgs a, int ctx_cap) {
  extern __shared__ __attribute__((aligned(16))) float sc_f32[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= a.B * a.heads) return;
  const int b = pair / a.heads, h = pair % a.heads;
  const int tgen = *a.step + 1, P1 = a.P1, ctx = P1 + tgen;
  float* sc = sc_f32 + (size_t)wave * ctx_cap;
  const float* kp = (const float*)a.kp + (size_t)h * P1 * 64;
  const float* vp = (const float*)a.vp + (size_t)h * P1 * 64;
  const size_t bh = (size_t)b * a.heads + h;
  const float* kc = (const float*)a.kc + bh * 8 * a.tmax * 8;
  const float* vc = (const float*)a.vc + bh * a.tmax * 64;
  const float* q = (const float*)a.q + (size_t)b * a.heads * 64 + h * 64;
  float qr[64];
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const float4 t = *(const float4*)(q + d);
    qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
  }
  float mx = -INFINITY;
  for (int j = lane; j < ctx; j += 64) {
    float acc = 0.f;
    if (j < P1) {
      const float* kr = kp + (size_t)j * 64;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const float4 t = *(const float4*)(kr + d);
        acc = fmaf(qr[d], t.x, acc); acc = fmaf(qr[d + 1], t.y, acc); acc = fmaf(qr[d + 2], t.z, acc); acc = fmaf(qr[d + 3], t.w, acc);
      }
    } else {
      const int t_ = j - P1;  // chunk-major own keys: element d of key t at ((d >> 3) * tmax + t) * 8 + (d & 7)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float* kr = kc + ((size_t)c * a.tmax + t_) * 8;
        const float4 t0 = *(const float4*)kr, t1 = *(const float4*)(kr + 4);
        acc = fmaf(qr[c * 8], t0.x, acc); acc = fmaf(qr[c * 8 + 1], t0.y, acc); acc = fmaf(qr[c * 8 + 2], t0.z, acc); acc = fmaf(qr[c * 8 + 3], t0.w, acc);
        acc = fmaf(qr[c * 8 + 4], t1.x, acc); acc = fmaf(qr[c * 8 + 5], t1.y, acc); acc = fmaf(qr[c * 8 + 6], t1.z, acc); acc = fmaf(qr[c * 8 + 7], t1.w, acc);
      }
    }
    sc[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < ctx; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float o = 0.f;  // output dimension `lane`
  for (int j = 0; j < P1; ++j) o = fmaf(sc[j], vp[(size_t)j * 64 + lane], o);
  for (int t_ = 0; t_ < tgen; ++t_) o = fmaf(sc[P1 + t_], vc[(size_t)t_ * 64 + lane], o);
  ((float*)a.out)[(size_t)b * a.heads * 64 + h * 64 + lane] = o / sum;
}

int decode_attn_f32_launch(const DecodeAttnArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.ngroups <= 1, "decode_attention (fp32 verification mode): one utterance per batch");
  const int ctx_cap = a.P1 + a.tmax;
  const size_t smem = (size_t)4 * ctx_cap * sizeof(float);
  TT_REQUIRE(smem <= 64 * 1024, "decode_attention (fp32 verification mode): context %d too long for the score buffer", ctx_cap);
  ProfScope ps(PROF_DECODE_ATTN, stream, 4.0 * a.B * a.heads * 64.0 * (a.P1 + a.host_tgen), ((double)a.B * a.host_tgen + a.P1) * a.heads * 64 * 2 * 4.0, true);
  launch_timed(ps, decode_attn_f32_kernel, dim3(cdiv(a.B * a.heads, 4)), dim3(256), smem, stream, a, ctx_cap);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
