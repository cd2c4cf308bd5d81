// GEMV-shaped decode GEMMs for handles that decode at most 4 sequences at a time (round 6; the streaming path, tortoise/api_fast.py:389-420,
// pulls ONE sequence token by token through tortoise/models/autoregressive.py:150-163).
//
// At one row the row tile of an MFMA kernel is padding: the 32 x 16 tile still moves 64 KB of activation rows (31 of 32 of them copies of
// row 0) next to 32 KB of weights per workgroup and pays the LDS ring's fill and barriers for 16 MFMA-steps of work.  Here a workgroup of 4
// waves owns 16 output columns, a wave 4 of them: it requests the W rows of its columns in full (K / 512 sixteen-byte loads per lane per
// row, ALL of them before the first use - one HBM round trip for the whole launch), holds the M <= 4 activation rows in registers, multiplies
// on the packed-pair dot instruction (v_dot2_f32_bf16 / v_dot2_f32_f16, f32 accumulation) and finishes every (row, column) with one
// cross-lane sum.  No LDS, no barrier, no split-K: the projections add bias + product onto the residual row in place, so the LayerNorm behind
// them has no slabs to fold.  The summation order is fixed (lane-local over k, then the DPP tree): deterministic, and a handle either always
// or never runs these kernels (the choice is the handle's max_batch, not the batch of a call), so chunked == one-shot decoding stays
// bit-exact.  They are NOT the bits of the MFMA path - a sharded job (per-rank batches of >= 16) never sees them.
// The two GEMVs behind a LayerNorm (QKV, c_fc) compute the norm themselves (template flag LN): five launches per layer, not seven.
// In situ (bench.py --workload stream, profiles/r06_ab_stream_gemv.txt): first chunk 62.5 ms (MFMA tiles) -> 56.0 (GEMV) -> 46.7 ms (norms inside).
// Measured against the product's skinny MFMA tile at M = 1 (scripts/kbench.py gemv, cold weights): QKV 5.04 -> 3.31 us, c_fc 5.15 -> 3.72,
// lm_head 6.83 -> 5.39; at M = 4 the two are equal, at M = 8 the MFMA tile wins - hence the <= 4 rule.
#include "ops.h"

namespace tt {

__device__ __forceinline__ float gv_dot8(Vec<bf16>::x8 a, Vec<bf16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}
__device__ __forceinline__ float gv_dot8(Vec<f16>::x8 a, Vec<f16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}

// EPI: GEMV_F32 out_f32 = acc + bias (lm_head) | GEMV_RES x += acc + bias in place (projections) | GEMV_GELU_T out_t = gelu_tanh(acc + bias) (c_fc)
//      | GEMV_QKV q * scale -> qbuf, K / V appended to the per-sequence cache at the device-side step (EpiQkvDecode's layout)
// LN: the activation rows are LayerNorm(ln_x rows) computed here (K == 1024: a lane's sixteen channels of each f32 residual row, two-pass variance
//     over the wave, affine, rounded to T like the row-norm kernel's output) while the weight rows are in flight - every workgroup repeats the
//     4 KB row's norm instead of a launch of its own doing it once.
template <typename T, int MR, int KC, int EPI, bool LN>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  typedef typename Vec<T>::x8 x8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16 + wave * 4;  // this wave's four columns (N % 4 == 0: whole quads, never ragged inside a wave)
  if (n0 >= a.N) return;
  const T* W = (const T*)a.W;
  const T* A = (const T*)a.A;
  x8 w[4][KC];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const T* wr = W + (size_t)(n0 + c) * a.ldw + lane * 8;
#pragma unroll
    for (int k = 0; k < KC; ++k) w[c][k] = *(const x8*)(wr + k * 512);
  }
  x8 x[MR][KC];
  if constexpr (LN) {
    static_assert(!LN || KC == 2, "the fused LayerNorm is the trunk's 1024-channel norm");
    float4 f[MR][4], g[4], b[4];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const float* xr = a.ln_x + (size_t)min(r, a.M - 1) * a.ldx + lane * 8;
      f[r][0] = *(const float4*)xr; f[r][1] = *(const float4*)(xr + 4); f[r][2] = *(const float4*)(xr + 512); f[r][3] = *(const float4*)(xr + 516);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane * 8 + (j >> 1) * 512 + (j & 1) * 4;
      g[j] = *(const float4*)(a.ln_g + c);
      b[j] = *(const float4*)(a.ln_b + c);
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += f[r][j].x + f[r][j].y + f[r][j].z + f[r][j].w;
      const float mean = wave_sum(sum) * (1.0f / 1024.0f);
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[r][j].x -= mean; f[r][j].y -= mean; f[r][j].z -= mean; f[r][j].w -= mean;
        sq += f[r][j].x * f[r][j].x + f[r][j].y * f[r][j].y + f[r][j].z * f[r][j].z + f[r][j].w * f[r][j].w;
      }
      const float var = wave_sum(sq) * (1.0f / 1024.0f);
      if (a.guard && blockIdx.x == 0 && threadIdx.x == 0 && r < a.M && !(var < INFINITY)) atomicAdd(a.guard, 1);  // NaN / inf in the row (norm.hip's guard)
      const float rstd = rsqrtf(var + a.ln_eps);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 lo = f[r][2 * k], hi = f[r][2 * k + 1];
        const typename Vec<T>::x4 p = pack4<T>(lo.x * rstd * g[2 * k].x + b[2 * k].x, lo.y * rstd * g[2 * k].y + b[2 * k].y, lo.z * rstd * g[2 * k].z + b[2 * k].z,
                                               lo.w * rstd * g[2 * k].w + b[2 * k].w);
        const typename Vec<T>::x4 q = pack4<T>(hi.x * rstd * g[2 * k + 1].x + b[2 * k + 1].x, hi.y * rstd * g[2 * k + 1].y + b[2 * k + 1].y,
                                               hi.z * rstd * g[2 * k + 1].z + b[2 * k + 1].z, hi.w * rstd * g[2 * k + 1].w + b[2 * k + 1].w);
        x[r][k] = __builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const T* ar = A + (size_t)min(r, a.M - 1) * a.lda + lane * 8;  // rows beyond M repeat the last row (never stored)
#pragma unroll
      for (int k = 0; k < KC; ++k) x[r][k] = *(const x8*)(ar + k * 512);
    }
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = *(const float4*)(a.bias + n0);
  int t = 0;
  if (EPI == GEMV_QKV) t = *a.step;
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) acc = gv_dot8(x[r][k], w[c][k], acc);
      v[c] = wave_sum(acc);
    }
    if (lane != 0 || r >= a.M) continue;
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    if (EPI == GEMV_F32) {
      *(float4*)(a.out_f32 + (size_t)r * a.ldo32 + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (EPI == GEMV_RES) {
      float4* xr = (float4*)(a.out_f32 + (size_t)r * a.ldo32 + n0);
      const float4 o = *xr;
      *xr = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
    } else if (EPI == GEMV_GELU_T) {
      *(typename Vec<T>::x4*)((T*)a.out_t + (size_t)r * a.ldot + n0) = pack4<T>(gelu_tanh(v[0]), gelu_tanh(v[1]), gelu_tanh(v[2]), gelu_tanh(v[3]));
    } else {  // GEMV_QKV: column n0 = part * dmodel + h * 64 + d, d a multiple of 4 (gemm_impl.h EpiQkvDecode::store)
      const int part = n0 / a.dmodel, cc = n0 - part * a.dmodel;
      const int h = cc >> 6, d = cc & 63;
      const size_t bh = (size_t)r * a.heads + h;
      if (part == 0) {
        *(typename Vec<T>::x4*)((T*)a.qbuf + (size_t)r * a.dmodel + cc) = pack4<T>(v[0] * a.q_scale, v[1] * a.q_scale, v[2] * a.q_scale, v[3] * a.q_scale);
      } else if (part == 1) {
        *(typename Vec<T>::x4*)((T*)a.kc + ((bh * 8 + (d >> 3)) * a.tmax + t) * 8 + (d & 7)) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else {
        *(typename Vec<T>::x4*)((T*)a.vc + (bh * a.tmax + t) * 64 + d) = pack4<T>(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

bool gemv_supported(int dtype, const GemvArgs& a) {
  const bool rows_ok = a.ln_x ? (a.K == 1024 && (a.ldx & 3) == 0 && ((size_t)a.ln_x & 15) == 0 && a.ln_g && a.ln_b && ((size_t)a.ln_g & 15) == 0 && ((size_t)a.ln_b & 15) == 0 &&
                                 (a.epi == GEMV_QKV || a.epi == GEMV_GELU_T))
                              : (a.A && (a.lda & 7) == 0 && ((size_t)a.A & 15) == 0);
  return (dtype == DT_BF16 || dtype == DT_F16) && a.M >= 1 && a.M <= 4 && (a.K == 1024 || a.K == 2048 || a.K == 4096) && (a.N & 3) == 0 && rows_ok &&
         (a.ldw & 7) == 0 && ((size_t)a.W & 15) == 0 && (!a.bias || ((size_t)a.bias & 15) == 0) &&
         (a.epi == GEMV_QKV ? (a.step && a.qbuf && a.kc && a.vc && a.dmodel % 64 == 0 && a.N == 3 * a.dmodel && a.heads * 64 == a.dmodel)
          : a.epi == GEMV_GELU_T ? (a.out_t && ((size_t)a.out_t & 7) == 0 && (a.ldot & 3) == 0)
                                 : (a.out_f32 && ((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0));
}

template <typename T, int MR, int KC>
static void gemv_dispatch_epi(const ProfScope& ps, const GemvArgs& a, hipStream_t s) {
  const dim3 grid(cdiv(a.N, 16));
  if constexpr (KC == 2) {
    if (a.ln_x) {  // the two GEMVs behind a LayerNorm (gemv_supported)
      if (a.epi == GEMV_GELU_T) launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_GELU_T, true>, grid, dim3(256), 0, s, a);
      else launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_QKV, true>, grid, dim3(256), 0, s, a);
      return;
    }
  }
  switch (a.epi) {
    case GEMV_F32: launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_F32, false>, grid, dim3(256), 0, s, a); break;
    case GEMV_RES: launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_RES, false>, grid, dim3(256), 0, s, a); break;
    case GEMV_GELU_T: launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_GELU_T, false>, grid, dim3(256), 0, s, a); break;
    default: launch_timed(ps, gemv_kernel<T, MR, KC, GEMV_QKV, false>, grid, dim3(256), 0, s, a); break;
  }
}
template <typename T, int MR>
static void gemv_dispatch_k(const ProfScope& ps, const GemvArgs& a, hipStream_t s) {
  if (a.K == 1024) gemv_dispatch_epi<T, MR, 2>(ps, a, s);
  else if (a.K == 2048) gemv_dispatch_epi<T, MR, 4>(ps, a, s);
  else gemv_dispatch_epi<T, MR, 8>(ps, a, s);
}
template <typename T>
static void gemv_dispatch_m(const ProfScope& ps, const GemvArgs& a, hipStream_t s) {
  if (a.M == 1) gemv_dispatch_k<T, 1>(ps, a, s);
  else if (a.M == 2) gemv_dispatch_k<T, 2>(ps, a, s);
  else gemv_dispatch_k<T, 4>(ps, a, s);
}

int gemv_launch(int dtype, const GemvArgs& a, hipStream_t stream) {
  TT_REQUIRE(gemv_supported(dtype, a), "gemv: unsupported problem (M=%d N=%d K=%d epi=%d dtype=%d): M <= 4, K in {1024, 2048, 4096}, N %% 4 == 0, aligned operands", a.M, a.N, a.K,
             a.epi, dtype);
  // algorithmic work: the weights once, the rows in and out
  ProfScope ps(PROF_GEMV, stream, 2.0 * a.M * a.N * a.K, 2.0 * a.N * a.K + 2.0 * a.M * a.K + 4.0 * a.M * a.N, true);
  if (dtype == DT_BF16) gemv_dispatch_m<bf16>(ps, a, stream);
  else gemv_dispatch_m<f16>(ps, a, stream);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
