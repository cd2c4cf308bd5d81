// UnivNet generator kernels (reference: tortoise/models/vocoder.py:104-222, 267-312).
// The sample-rate side of the vocoder is 32 channels wide: too narrow for MFMA tiles, fp32 in the
// reference, and HBM/LDS-bound.  It runs as fp32 VALU kernels over channels-first [C][T] rows
// (coalesced along T).  The mel-rate KernelPredictor convolutions (64 -> 24576 channels) are the
// only GEMM-shaped part and go through the MFMA conv-GEMM (gemm.hip).
#include "ops.h"

namespace tt {

// ---------------------------------------------------------------- direct conv1d, thread per output sample
template <int COUT>
__global__ __launch_bounds__(256) void conv1d_direct_kernel(Conv1dArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cin][k][COUT]
  const int nw = a.Cin * a.k * COUT;
  for (int i = threadIdx.x; i < nw; i += 256) {
    const int co = i % COUT, kk = (i / COUT) % a.k, ci = i / (COUT * a.k);
    wl[i] = a.w[((size_t)co * a.Cin + ci) * a.k + kk];
  }
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.T) return;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = a.bias ? a.bias[co] : 0.f;
  const int half = a.k >> 1;
  for (int kk = 0; kk < a.k; ++kk) {
    int tt_ = t + (kk - half) * a.dilation;
    if (a.reflect) {
      if (tt_ < 0) tt_ = -tt_;
      if (tt_ >= a.T) tt_ = 2 * (a.T - 1) - tt_;
    } else if (tt_ < 0 || tt_ >= a.T) {
      continue;
    }
    for (int ci = 0; ci < a.Cin; ++ci) {
      float xv = a.x[(size_t)ci * a.T + tt_];
      if (a.in_slope >= 0.f) xv = xv > 0.f ? xv : xv * a.in_slope;
      const float* wr = wl + (ci * a.k + kk) * COUT;
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] += xv * wr[co];
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float v = acc[co];
    if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
    else if (a.out_act == 5) v = tanhf(v);
    a.y[(size_t)co * a.T + t] = v;
  }
}

// ---------------------------------------------------------------- 32 -> 32 channel k3 dilated conv1d on the f32 matrix cores (round 5)
// y[co][t] = act(b[co] + sum_{ci,tap} w[co][ci][tap] * lrelu(x[ci][t + (tap - 1) dil])) is the GEMM  W[32][96] . X_unf[96][T]  with
// kidx = 3 ci + tap - exactly the weight's own [Cout][Cin][k] row.  v_mfma_f32_32x32x2_f32 multiplies f32 operands exactly (the f32
// vector rate, 64 FLOP / clk / SIMD) but takes both operands from REGISTERS, one LDS read each per 4096 flops: the thread-per-sample
// kernel above issues one broadcast LDS weight read per FMA and ran at ~10 % of the vector rate (45 us per launch at T = 225 280).
// One workgroup = 128 samples: the input window (128 + 2 dil columns, LeakyReLU applied, zeros outside [0, T)) and the transposed
// weights sit in LDS; wave w owns samples 32 w .. 32 w + 31, 48 MFMA steps of two k each.
constexpr int CVM_TS = 128, CVM_MAXD = 27, CVM_ROW = CVM_TS + 2 * CVM_MAXD + 2;
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(Conv1dArgs a) {
  __shared__ float xs[32][CVM_ROW];
  __shared__ float wl[96][33];  // [kidx][co] (+1: the transposing fill is conflict-free)
  const int t0 = blockIdx.x * CVM_TS, d = a.dilation, width = CVM_TS + 2 * d;
  for (int i = threadIdx.x; i < 32 * 96; i += 256) {
    const int co = i / 96, kidx = i - co * 96;
    wl[kidx][co] = a.w[i];
  }
  for (int ci = threadIdx.x >> 6; ci < 32; ci += 4) {  // a wave per input row: coalesced
    for (int i = threadIdx.x & 63; i < width; i += 64) {
      const int t = t0 - d + i;
      float v = (t >= 0 && t < a.T) ? a.x[(size_t)ci * a.T + t] : 0.f;
      if (a.in_slope >= 0.f) v = v > 0.f ? v : v * a.in_slope;
      xs[ci][i] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int ci = 0, tap = h;  // kidx = 2 p + h = 3 ci + tap
  const float* xb = &xs[0][wave * 32 + j];
#pragma unroll 8
  for (int p = 0; p < 48; ++p) {
    const float av = wl[2 * p + h][j];
    const float bv = xb[ci * CVM_ROW + tap * d];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    tap += 2;
    if (tap >= 3) { tap -= 3; ++ci; }
  }
  const int t = t0 + wave * 32 + j;
  if (t < a.T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[r] + (a.bias ? a.bias[co] : 0.f);
      if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
      else if (a.out_act == 5) v = tanhf(v);
      a.y[(size_t)co * a.T + t] = v;
    }
  }
}

bool g_voc_mfma = true;  // tt_voc_variant: 0 = the thread-per-sample VALU kernels (A/B runs)

int conv1d_direct_launch(const Conv1dArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.Cout == 32 || a.Cout == 1, "conv1d_direct: Cout=%d unsupported (32 or 1)", a.Cout);
  if (g_voc_mfma && a.Cin == 32 && a.Cout == 32 && a.k == 3 && !a.reflect && a.dilation >= 1 && a.dilation <= CVM_MAXD) {
    ProfScope ps(PROF_CONV1D, stream, 2.0 * a.Cin * a.Cout * a.k * (double)a.T, 4.0 * (a.Cin + a.Cout) * (double)a.T);
    conv1d_mfma_kernel<<<cdiv(a.T, CVM_TS), 256, 0, stream>>>(a);
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const size_t smem = (size_t)a.Cin * a.k * a.Cout * sizeof(float);
  TT_REQUIRE(smem <= 60 * 1024, "conv1d_direct: weights do not fit LDS");
  const int blocks = cdiv(a.T, 256);
  ProfScope ps(PROF_CONV1D, stream, 2.0 * a.Cin * a.Cout * a.k * (double)a.T, 4.0 * (a.Cin + a.Cout) * (double)a.T);
  if (a.Cout == 32) conv1d_direct_kernel<32><<<blocks, 256, smem, stream>>>(a);
  else conv1d_direct_kernel<1><<<blocks, 256, smem, stream>>>(a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- ConvTranspose1d(C->C, k=2s, stride s, pad s/2)
// Output sample t = s*j + r - p receives x[:, j] * w[:, :, r] + x[:, j-1] * w[:, :, r+s]; one block
// column per phase r keeps both weight slices (2 x C x C) in LDS.
__global__ __launch_bounds__(256) void convt1d_kernel(ConvT1dArgs a) {
  __shared__ float w0[32 * 32], w1[32 * 32];  // [ci][co]
  const int C = a.C, s = a.stride, r = blockIdx.y;
  const int p = s / 2 + s % 2;
  for (int i = threadIdx.x; i < C * C; i += 256) {
    const int ci = i / C, co = i % C;
    w0[i] = a.w[((size_t)ci * C + co) * (2 * s) + r];
    w1[i] = a.w[((size_t)ci * C + co) * (2 * s) + r + s];
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int Tout = a.Tin * s;
  const int t = s * j + r - p;
  if (j > a.Tin || t < 0 || t >= Tout) return;
  float acc[32];
#pragma unroll
  for (int co = 0; co < 32; ++co) acc[co] = co < C ? a.bias[co] : 0.f;
  for (int ci = 0; ci < C; ++ci) {
    float x0 = j < a.Tin ? a.x[(size_t)ci * a.Tin + j] : 0.f;
    float x1 = j >= 1 ? a.x[(size_t)ci * a.Tin + j - 1] : 0.f;
    if (a.in_slope >= 0.f) {
      x0 = x0 > 0.f ? x0 : x0 * a.in_slope;
      x1 = x1 > 0.f ? x1 : x1 * a.in_slope;
    }
#pragma unroll
    for (int co = 0; co < 32; ++co) acc[co] += x0 * w0[ci * C + co] + x1 * w1[ci * C + co];
  }
  for (int co = 0; co < C; ++co) a.y[(size_t)co * Tout + t] = acc[co];
}
int convt1d_launch(const ConvT1dArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.C == 32 && a.stride % 2 == 0, "convt1d: C=%d stride=%d unsupported (C == 32, even stride)", a.C, a.stride);
  dim3 grid(cdiv(a.Tin + 1, 256), a.stride);
  ProfScope ps(PROF_CONVT, stream, 2.0 * a.C * a.C * 2.0 * a.Tin * a.stride, 4.0 * a.C * (double)a.Tin * (1 + a.stride));
  convt1d_kernel<<<grid, 256, 0, stream>>>(a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- location-variable convolution + gate
// One frame's predicted kernel [6144] in the operand type -> f32 in LDS; returns true when a non-finite tap was staged.
template <typename T>
__device__ __forceinline__ bool lvc_stage_kernel(const void* kernels, size_t elem_off, float* wk) {
  bool bad = false;
  if constexpr (sizeof(T) == 4) {
    const float* kg = (const float*)kernels + elem_off;
    for (int i = threadIdx.x * 4; i < 32 * 64 * 3; i += 1024) {
      const float4 kv = *(const float4*)(kg + i);
      bad = bad || !(fabsf(kv.x) < INFINITY) || !(fabsf(kv.y) < INFINITY) || !(fabsf(kv.z) < INFINITY) || !(fabsf(kv.w) < INFINITY);
      *(float4*)(wk + i) = kv;
    }
  } else {
    typedef typename Vec<T>::x8 x8;
    const T* kg = (const T*)kernels + elem_off;
    for (int i = threadIdx.x * 8; i < 32 * 64 * 3; i += 2048) {
      const x8 kv = *(const x8*)(kg + i);
      float f[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        f[c] = (float)kv[c];
        bad = bad || !(fabsf(f[c]) < INFINITY);
      }
      *(float4*)(wk + i) = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)(wk + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
  }
  return bad;
}

// vocoder.py:182-216 (dilation 1) fused with the sigmoid*tanh gate and the residual add (vocoder.py:178-179).
// One block per mel frame l: its [32][64][3] kernel and the (hop+2)-sample input window sit in LDS.
template <typename KT, int HOP>
__global__ __launch_bounds__(256) void lvc_kernel(LvcArgs a) {
  constexpr int OG = 256 / HOP;     // output groups across threads
  constexpr int HALF = 32 / OG;     // gate pairs per thread
  __shared__ __attribute__((aligned(16))) float wk[32 * 64 * 3];
  __shared__ float xs[32][HOP + 2];
  const int l = blockIdx.x;
  const int T = a.L * HOP;
  const bool bad = lvc_stage_kernel<KT>(a.kernels, (size_t)l * a.ldk + a.koff, wk);
  // an overflowed fp16 KernelPredictor operand shows up HERE as inf / NaN taps; behind the sigmoid * tanh gate it would be a finite sample
  if (a.guard && __any(bad) && (threadIdx.x & 63) == 0) atomicAdd(a.guard, 1);
  for (int i = threadIdx.x; i < 32 * (HOP + 2); i += 256) {
    const int ci = i / (HOP + 2), off = i % (HOP + 2);
    const int t = l * HOP + off - 1;
    float v = (t >= 0 && t < T) ? a.x_in[(size_t)ci * T + t] : 0.f;
    if (a.in_slope >= 0.f) v = v > 0.f ? v : v * a.in_slope;
    xs[ci][off] = v;
  }
  __syncthreads();
  const int s = threadIdx.x % HOP, og = threadIdx.x / HOP;
  float acc[2 * HALF];
  const float* bl = a.bias + (size_t)l * a.ldb + a.boff;
#pragma unroll
  for (int u = 0; u < HALF; ++u) {
    acc[u] = bl[og * HALF + u];
    acc[HALF + u] = bl[32 + og * HALF + u];
  }
  for (int ci = 0; ci < 32; ++ci) {
    const float x0 = xs[ci][s], x1 = xs[ci][s + 1], x2 = xs[ci][s + 2];
    const float* w = wk + ci * 192;
#pragma unroll
    for (int u = 0; u < HALF; ++u) {
      const float* wa = w + (og * HALF + u) * 3;
      const float* wg = w + (32 + og * HALF + u) * 3;
      acc[u] += x0 * wa[0] + x1 * wa[1] + x2 * wa[2];
      acc[HALF + u] += x0 * wg[0] + x1 * wg[1] + x2 * wg[2];
    }
  }
  const int t = l * HOP + s;
#pragma unroll
  for (int u = 0; u < HALF; ++u) {
    const int o = og * HALF + u;
    const float g = 1.f / (1.f + expf(-acc[u]));
    a.x[(size_t)o * T + t] += g * tanhf(acc[HALF + u]);
  }
}
// The same on the f32 matrix cores (round 5; hops of 64 and 256 samples): per frame, out[64][HOP] = K_l[64][96] . X_unf[96][HOP] with
// kidx = 3 ci + tap; K_l is the frame's predicted kernel as the KernelPredictor GEMM left it ([ci][co][tap]: co is the MFMA row, read at a
// stride of 3 floats - conflict-free).  A wave owns sample blocks of 32 and BOTH output halves of them (sigmoid half rows 0 .. 31, tanh
// half rows 32 .. 63 land in the same lane / register of two accumulators), so the gate needs no exchange.
template <typename KT, int HOP>
__global__ __launch_bounds__(256) void lvc_mfma_kernel(LvcArgs a) {
  constexpr int NSB = HOP / 32;                    // sample blocks per frame
  constexpr int SBW = NSB >= 4 ? NSB / 4 : 1;      // sample blocks per wave (waves beyond NSB idle after the staging)
  __shared__ __attribute__((aligned(16))) float wk[32 * 64 * 3];
  __shared__ float xs[32][HOP + 2];
  const int l = blockIdx.x;
  const int T = a.L * HOP;
  const bool bad = lvc_stage_kernel<KT>(a.kernels, (size_t)l * a.ldk + a.koff, wk);
  // an overflowed fp16 KernelPredictor operand shows up HERE as inf / NaN taps; behind the sigmoid * tanh gate it would be a finite sample
  if (a.guard && __any(bad) && (threadIdx.x & 63) == 0) atomicAdd(a.guard, 1);
  for (int i = threadIdx.x; i < 32 * (HOP + 2); i += 256) {
    const int ci = i / (HOP + 2), off = i % (HOP + 2);
    const int t = l * HOP + off - 1;
    float v = (t >= 0 && t < T) ? a.x_in[(size_t)ci * T + t] : 0.f;
    if (a.in_slope >= 0.f) v = v > 0.f ? v : v * a.in_slope;
    xs[ci][off] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave * SBW >= NSB) return;
  const int j = lane & 31, h = lane >> 5;
  f32x16 acc[SBW][2];
#pragma unroll
  for (int u = 0; u < SBW; ++u)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][c][r] = 0.f;
  int ci = 0, tap = h;
#pragma unroll 4
  for (int p = 0; p < 48; ++p) {
    const float a0 = wk[ci * 192 + j * 3 + tap], a1 = wk[ci * 192 + (32 + j) * 3 + tap];
#pragma unroll
    for (int u = 0; u < SBW; ++u) {
      const float bv = xs[ci][(wave * SBW + u) * 32 + j + tap];
      acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[u][0], 0, 0, 0);
      acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[u][1], 0, 0, 0);
    }
    tap += 2;
    if (tap >= 3) { tap -= 3; ++ci; }
  }
  const float* bl = a.bias + (size_t)l * a.ldb + a.boff;
#pragma unroll
  for (int u = 0; u < SBW; ++u) {
    const int t = l * HOP + (wave * SBW + u) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float g = 1.f / (1.f + expf(-(acc[u][0][r] + bl[o])));
      a.x[(size_t)o * T + t] += g * tanhf(acc[u][1][r] + bl[32 + o]);
    }
  }
}

int lvc_launch(const LvcArgs& a, hipStream_t stream) {
  const int eb = a.dtype == DT_F32 ? 4 : 2;
  const int al = 16 / eb;  // elements per 16-byte staging load
  TT_REQUIRE(a.dtype == DT_BF16 || a.dtype == DT_F16 || a.dtype == DT_F32, "lvc: unknown kernel dtype %d", a.dtype);
  TT_REQUIRE(a.ldk % al == 0 && a.koff % al == 0 && ((size_t)a.kernels & 15) == 0, "lvc: kernel rows must be 16-byte aligned");
  // algorithmic bytes: the predicted kernels (L x 6144, operand type) are read once, x_in read + x updated
  ProfScope ps(PROF_LVC, stream, 2.0 * 96 * 64 * (double)a.L * a.hop, (double)eb * 6144.0 * a.L + 4.0 * 64 * a.L + 4.0 * 96 * (double)a.L * a.hop);
  const bool mfma = g_voc_mfma && a.in_slope < 0.f && (a.hop == 64 || a.hop == 256);
#define TT_LVC(T)                                                                  \
  do {                                                                             \
    if (mfma && a.hop == 64) lvc_mfma_kernel<T, 64><<<a.L, 256, 0, stream>>>(a);   \
    else if (mfma) lvc_mfma_kernel<T, 256><<<a.L, 256, 0, stream>>>(a);            \
    else if (a.hop == 8) lvc_kernel<T, 8><<<a.L, 256, 0, stream>>>(a);             \
    else if (a.hop == 64) lvc_kernel<T, 64><<<a.L, 256, 0, stream>>>(a);           \
    else if (a.hop == 256) lvc_kernel<T, 256><<<a.L, 256, 0, stream>>>(a);         \
    else { set_error("lvc: hop=%d unsupported (8, 64, 256)", a.hop); return -1; }  \
  } while (0)
  if (a.dtype == DT_F32) TT_LVC(float);
  else if (a.dtype == DT_BF16) TT_LVC(bf16);
  else TT_LVC(f16);
#undef TT_LVC
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
