// Small HBM-bound glue kernels: rotary, gathers, casts, the CLVP score tail and the fused
// diffusion sampler epilogue (classifier-free guidance + learned-range variance + posterior mean +
// noise + operand re-quantisation for the next step, one pass over the 100 x S state).
#include "ops.h"

namespace tt {

// x-transformers apply_rotary_pos_emb on the leading `rot` dims (xtransformers.py:277-286, 625-629):
// t[d] = t[d]*cos(theta_d) - t[d+rot/2]*sin ; t[d+rot/2] = t[d+rot/2]*cos + t[d]*sin, theta = s * inv_freq[d]
// Two passes with their own thread orders so that both are coalesced: q / k rows are [n][64] (dimension fastest), V^T rows are
// [64][n_pad] (position fastest).  Round 3's single pass walked V^T with a stride of n_pad elements between neighbouring lanes and
// called sincosf per element: 163 us per launch on CLVP's speech tower (12 heads x 256 candidates x 200 codes), 3.3 ms per utterance.
template <typename T>
__global__ void rotary_qk_kernel(T* q, T* k, const float* inv_freq, long rows, int n, int rot) {
  const int half = rot >> 1;
  const long total = rows * half;  // rows = BH * n
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int d = (int)(f % half);
    const long row = f / half;
    const int s = (int)(row % n);
    float sn, cs;
    sincosf((float)s * inv_freq[d], &sn, &cs);
    T* qr = q + row * 64;
    T* kr = k + row * 64;
    const float qa = (float)qr[d], qb = (float)qr[d + half], ka = (float)kr[d], kb = (float)kr[d + half];
    qr[d] = (T)(qa * cs - qb * sn);
    qr[d + half] = (T)(qb * cs + qa * sn);
    kr[d] = (T)(ka * cs - kb * sn);
    kr[d + half] = (T)(kb * cs + ka * sn);
  }
}
template <typename T>
__global__ void rotary_vt_kernel(T* vt, const float* inv_freq, long BH, int n, int n_pad, int rot) {
  const int half = rot >> 1;
  const long total = BH * half * n;  // position fastest: neighbouring lanes touch neighbouring elements of a V^T row
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int s = (int)(f % n);
    const int d = (int)((f / n) % half);
    const long bh = f / ((long)n * half);
    float sn, cs;
    sincosf((float)s * inv_freq[d], &sn, &cs);
    T* v0 = vt + (bh * 64 + d) * n_pad + s;
    T* v1 = vt + (bh * 64 + d + half) * n_pad + s;
    const float a = (float)*v0, b = (float)*v1;
    *v0 = (T)(a * cs - b * sn);
    *v1 = (T)(b * cs + a * sn);
  }
}
int rotary_launch(int dtype, void* q, void* k, void* vt, const float* inv_freq, int BH, int n, int n_pad, int rot,
                  hipStream_t stream) {
  TT_REQUIRE(rot > 0 && rot <= 64 && rot % 2 == 0, "rotary: bad rot=%d", rot);
  const long total = (long)BH * n * (rot / 2);
  const int blocks = (int)std::min<long>(cdiv64(total, 256), 16384);
  TT_DISPATCH_T(dtype, T, rotary_qk_kernel<T><<<blocks, 256, 0, stream>>>((T*)q, (T*)k, inv_freq, (long)BH * n, n, rot));
  TT_CHECK_HIP(hipGetLastError());
  TT_DISPATCH_T(dtype, T, rotary_vt_kernel<T><<<blocks, 256, 0, stream>>>((T*)vt, inv_freq, (long)BH, n, n_pad, rot));
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void gather_rows_kernel(const float* src, const int* idx, float* dst, int rows, int C) {
  const int c4n = C >> 2;
  const long total = (long)rows * c4n;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int r = (int)(f / c4n), c = (int)(f % c4n) * 4;
    *(float4*)(dst + (size_t)r * C + c) = *(const float4*)(src + (size_t)idx[r] * C + c);
  }
}
int gather_rows_launch(const float* src, const int* idx, float* dst, int rows, int C, hipStream_t stream) {
  TT_REQUIRE(C % 4 == 0, "gather_rows: C must be a multiple of 4");
  const int blocks = (int)std::min<long>(cdiv64((long)rows * (C / 4), 256), 4096);
  gather_rows_kernel<<<blocks, 256, 0, stream>>>(src, idx, dst, rows, C);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void mean_rows_kernel(const float* src, float* dst, int n, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < n; ++r) s += src[((size_t)b * n + r) * C + c];
  dst[(size_t)b * C + c] = s / (float)n;
}
int mean_rows_launch(const float* src, float* dst, int B, int n, int C, hipStream_t stream) {
  dim3 grid(cdiv(C, 256), B);
  mean_rows_kernel<<<grid, 256, 0, stream>>>(src, dst, n, C);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// clvp.py:129-135: normalize both latents (F.normalize eps 1e-12), row-wise dot, * exp(temperature)
__global__ __launch_bounds__(256) void clvp_score_kernel(const float* t, int t_rows, const float* s, const float* temperature,
                                                         float* out, int D) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* tr = t + (size_t)(t_rows == 1 ? 0 : b) * D;
  const float* sr = s + (size_t)b * D;
  float tt_ = 0.f, ss = 0.f, ts = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    const float x = tr[c], y = sr[c];
    tt_ += x * x;
    ss += y * y;
    ts += x * y;
  }
  tt_ = block_sum_256(tt_, red);
  ss = block_sum_256(ss, red);
  ts = block_sum_256(ts, red);
  if (threadIdx.x == 0) {
    const float nt = fmaxf(sqrtf(tt_), 1e-12f), ns = fmaxf(sqrtf(ss), 1e-12f);
    out[b] = ts / (nt * ns) * expf(*temperature);
  }
}
int clvp_score_launch(const float* t, int t_rows, const float* s, const float* temperature, float* out, int B, int D,
                      hipStream_t stream) {
  clvp_score_kernel<<<B, 256, 0, stream>>>(t, t_rows, s, temperature, out, D);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ void cast_pad_kernel(const float* src, int lds, T* dst, int ldd, int rows, int c, int cpad) {
  const long total = (long)rows * cpad;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int r = (int)(f / cpad), j = (int)(f % cpad);
    dst[(size_t)r * ldd + j] = j < c ? (T)src[(size_t)r * lds + j] : (T)0.f;
  }
}
int cast_pad_launch(int dtype, const float* src, int lds, void* dst, int ldd, int rows, int c, int cpad, hipStream_t stream) {
  const int blocks = (int)std::min<long>(cdiv64((long)rows * cpad, 256), 4096);
  TT_DISPATCH_T(dtype, T, cast_pad_kernel<T><<<blocks, 256, 0, stream>>>(src, lds, (T*)dst, ldd, rows, c, cpad));
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void broadcast_rows_kernel(const float* vec, float* dst, int rows, int C) {
  const long total = (long)rows * C;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) dst[f] = vec[f % C];
}
__global__ void repeat_rows_kernel(const float4* src, float4* dst, size_t n4, int reps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    for (int j = 0; j < reps; ++j) dst[(size_t)j * n4 + i] = v;
  }
}
int repeat_rows_launch(const float* src, float* dst, int rows, int reps, int C, hipStream_t stream) {
  TT_REQUIRE(C % 4 == 0, "repeat_rows: C must be a multiple of 4");
  const size_t n4 = (size_t)rows * C / 4;
  const int blocks = (int)std::min<long>(cdiv64((long)n4, 256), 4096);
  repeat_rows_kernel<<<blocks, 256, 0, stream>>>((const float4*)src, (float4*)dst, n4, reps);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

int broadcast_rows_launch(const float* vec, float* dst, int rows, int C, hipStream_t stream) {
  const int blocks = (int)std::min<long>(cdiv64((long)rows * C, 256), 4096);
  broadcast_rows_kernel<<<blocks, 256, 0, stream>>>(vec, dst, rows, C);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void transpose_kernel(const float* src, float* dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][i];
  }
}
int transpose_launch(const float* src, float* dst, int rows, int cols, hipStream_t stream) {
  dim3 grid(cdiv(cols, 32), cdiv(rows, 32));
  transpose_kernel<<<grid, 256, 0, stream>>>(src, dst, rows, cols);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ void silu_cast_kernel(const float* src, T* dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (T)silu(src[i]);
}
int silu_cast_launch(int dtype, const float* src, void* dst, int n, hipStream_t stream) {
  const int blocks = std::min(cdiv(n, 256), 4096);
  TT_DISPATCH_T(dtype, T, silu_cast_kernel<T><<<blocks, 256, 0, stream>>>(src, (T*)dst, n));
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// GaussianDiffusion.p_mean_variance + p_sample (utils/diffusion.py:312-418, 487-531) for one spaced
// index.  Thread per (s, c).  Model output rows are token-major [S][2C]: eps = cols 0..C-1, var = C..2C-1.
template <typename T>
__global__ void psample_kernel(PSampleArgs a) {
  const int S = a.S, C = a.C;
  const long total = (long)S * a.cpad;
  const int ld = a.ld_rows > 0 ? a.ld_rows : S;  // rows between the conditioned and the conditioning-free batch row
  const float* oc = a.out;
  const float* ou = a.out + (size_t)ld * 2 * C;
  const int slot = *a.slot;
  const PSampleStep st = a.steps[slot];
  const float* noise_base = a.io ? (const float*)a.io[0] : a.noise;
  float* mel_out = a.io ? (float*)a.io[1] : a.mel_out;
  const float* noise = noise_base ? noise_base + (size_t)slot * C * S : nullptr;
  bool bad = false;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int s = (int)(f / a.cpad), c = (int)(f % a.cpad);
    T xt = (T)0.f;
    if (c < C) {
      const float x = a.x[(size_t)s * C + c];
      float eps = oc[(size_t)s * 2 * C + c];
      const float var = oc[(size_t)s * 2 * C + C + c];
      bad = bad || !(fabsf(eps) < INFINITY) || !(fabsf(var) < INFINITY);
      const float frac = (var + 1.f) * 0.5f;
      const float log_var = frac * st.max_log + (1.f - frac) * st.min_log;
      if (a.has_uncond) {
        const float eu = ou[(size_t)s * 2 * C + c];
        bad = bad || !(fabsf(eu) < INFINITY);
        eps = (1.f + st.cfk) * eps - st.cfk * eu;
      }
      float x0 = st.sqrt_recip * x - st.sqrt_recipm1 * eps;
      x0 = fminf(1.f, fmaxf(-1.f, x0));
      float xn = st.coef1 * x0 + st.coef2 * x;
      if (st.nonzero != 0.f && noise) xn += st.nonzero * expf(0.5f * log_var) * noise[(size_t)c * S + s];
      a.x[(size_t)s * C + c] = xn;
      if (mel_out) mel_out[(size_t)c * S + s] = (xn + 1.f) * 0.5f * a.mel_scale + a.mel_shift;
      xt = (T)xn;
    }
    if (a.x_t) {
      T* d = (T*)a.x_t;
      d[(size_t)s * a.cpad + c] = xt;
      if (a.has_uncond) d[((size_t)ld + s) * a.cpad + c] = xt;  // the conditioning-free batch row reads the same state
    }
  }
  // the x0 clamp (fminf / fmaxf drop a NaN) would hide a non-finite model output: count it here
  if (a.guard && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(a.guard, 1);
}
// End of a sampler step: advance the device-side step counter and stage the NEXT step's scale / shift rows at a fixed
// address (ss_cur), so that every GroupNorm of the next step reads them with plain up-front loads instead of a dependent
// "load the counter, then the row it selects" chain (two extra memory round trips per GroupNorm launch).  One block: the
// counter is read by every thread before it is bumped.
__global__ __launch_bounds__(1024) void slot_advance_kernel(int* slot, const float* ss_all, float* ss_cur, int row_floats, int last_slot) {
  const int nxt = min(*slot + 1, last_slot);
  if (ss_all) {
    const float4* src = (const float4*)(ss_all + (size_t)nxt * row_floats);
    float4* dst = (float4*)ss_cur;
    for (int i = threadIdx.x; i < row_floats / 4; i += 1024) dst[i] = src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) *slot += 1;
}
int slot_advance_launch(int* slot, const float* ss_all, float* ss_cur, int row_floats, int last_slot, hipStream_t stream) {
  TT_REQUIRE(row_floats % 4 == 0, "slot_advance: row size must be a multiple of 4 floats");
  slot_advance_kernel<<<1, 1024, 0, stream>>>(slot, ss_all, ss_cur, row_floats, last_slot);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}
int psample_launch(int dtype, const PSampleArgs& a, hipStream_t stream) {
  const int blocks = (int)std::min<long>(cdiv64((long)a.S * a.cpad, 256), 4096);
  TT_DISPATCH_T(dtype, T, psample_kernel<T><<<blocks, 256, 0, stream>>>(a));
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
