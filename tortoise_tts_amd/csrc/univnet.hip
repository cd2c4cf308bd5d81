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

int conv1d_direct_launch(const Conv1dArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.Cout == 32 || a.Cout == 1, "conv1d_direct: Cout=%d unsupported (32 or 1)", a.Cout);
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
// vocoder.py:182-216 (dilation 1) fused with the sigmoid*tanh gate and the residual add (vocoder.py:178-179).
// One block per mel frame l: its [32][64][3] kernel and the (hop+2)-sample input window sit in LDS.
template <int HOP>
__global__ __launch_bounds__(256) void lvc_kernel(LvcArgs a) {
  constexpr int OG = 256 / HOP;     // output groups across threads
  constexpr int HALF = 32 / OG;     // gate pairs per thread
  __shared__ __attribute__((aligned(16))) float wk[32 * 64 * 3];
  __shared__ float xs[32][HOP + 2];
  const int l = blockIdx.x;
  const int T = a.L * HOP;
  const float* kg = a.kernels + (size_t)l * a.ldk + a.koff;
  for (int i = threadIdx.x * 4; i < 32 * 64 * 3; i += 1024) *(float4*)(wk + i) = *(const float4*)(kg + i);
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
int lvc_launch(const LvcArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.ldk % 4 == 0 && a.koff % 4 == 0, "lvc: kernel rows must be 16-byte aligned");
  // algorithmic bytes: the predicted kernels (L x 6144 f32) are read once, x_in read + x updated
  ProfScope ps(PROF_LVC, stream, 2.0 * 96 * 64 * (double)a.L * a.hop, 4.0 * (6144.0 + 64) * a.L + 4.0 * 96 * (double)a.L * a.hop);
  switch (a.hop) {
    case 8: lvc_kernel<8><<<a.L, 256, 0, stream>>>(a); break;
    case 64: lvc_kernel<64><<<a.L, 256, 0, stream>>>(a); break;
    case 256: lvc_kernel<256><<<a.L, 256, 0, stream>>>(a); break;
    default: set_error("lvc: hop=%d unsupported (8, 64, 256)", a.hop); return -1;
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
