// Row norms (LayerNorm / x-transformers RMSNorm) fused with the residual-stream update, and
// token-major GroupNorm (stats + apply) for DiffusionTts.  All statistics are fp32 (GroupNorm's
// cross-chunk combine is fp64); normalised activations are emitted directly in the GEMM operand
// type so no separate cast pass exists anywhere in the engine.
#include "ops.h"

namespace tt {

// ------------------------------------------------------------------------------- row norm
// One 256-thread block per row, D <= 4096, D % 4 == 0.
template <typename T, int NSLAB>
__global__ __launch_bounds__(256) void rownorm_kernel(RowNormArgs a) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  constexpr int J = 4;
  float4 v[J];
  float* xr = a.x + (size_t)row * a.ldx;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = (tid + 256 * j) * 4;
    if (c < a.D) {
      float4 t = a.x_in ? *(const float4*)(a.x_in + (size_t)row * a.ldxin + c) : *(const float4*)(xr + c);
      if (a.add_bias) {
        const float4 b = *(const float4*)(a.add_bias + c);
        t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
      }
      if constexpr (NSLAB >= 0) {
        float4 sl[NSLAB > 0 ? NSLAB : 1];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)  // all partial-sum slabs requested at once, summed in slab order
          sl[s] = *(const float4*)(a.add_slabs + (size_t)s * a.slab_stride + (size_t)row * a.ldslab + c);
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
          t.x += sl[s].x; t.y += sl[s].y; t.z += sl[s].z; t.w += sl[s].w;
        }
      } else {  // odd slab counts: same order, one round trip per slab
        for (int s = 0; s < a.nslab; ++s) {
          const float4 p = *(const float4*)(a.add_slabs + (size_t)s * a.slab_stride + (size_t)row * a.ldslab + c);
          t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
      }
      if (a.write_x) *(float4*)(xr + c) = t;
      v[j] = t;
      sum += t.x + t.y + t.z + t.w;
    } else {
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (a.mode == NORM_NONE) return;

  float* o32 = a.out_f32;
  if (o32 && a.f32_slot) o32 += (size_t)(*a.f32_slot + a.f32_slot_base) * a.f32_slot_stride;
  auto emit = [&](const float4* y) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = (tid + 256 * j) * 4;
      if (c < a.D) {
        if (a.out_t) *(typename Vec<T>::x4*)((T*)a.out_t + (size_t)row * a.ldot + c) = pack4<T>(y[j].x, y[j].y, y[j].z, y[j].w);
        if (o32) *(float4*)(o32 + (size_t)row * a.ldo32 + c) = y[j];
      }
    }
  };

  float4 y[J];
  if (a.mode == NORM_RMS) {
    // x-transformers RMSNorm: x / max(||x|| * D^-0.5, eps) * g
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) sq += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    sq = block_sum_256(sq, red);
    if (a.guard && tid == 0 && !(sq < INFINITY)) atomicAdd(a.guard, 1);  // NaN / inf in the row: an operand overflowed upstream
    const float nrm = sqrtf(sq) * rsqrtf((float)a.D);
    const float inv = 1.0f / fmaxf(nrm, a.eps1);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = (tid + 256 * j) * 4;
      if (c < a.D) {
        const float4 g = *(const float4*)(a.g1 + c);
        y[j] = make_float4(v[j].x * inv * g.x, v[j].y * inv * g.y, v[j].z * inv * g.z, v[j].w * inv * g.w);
      }
    }
    emit(y);
    return;
  }

  // LayerNorm (two-pass variance in registers), optionally followed by a second LayerNorm.
  const float* gs[2] = {a.g1, a.g2};
  const float* bs[2] = {a.b1, a.b2};
  const float epss[2] = {a.eps1, a.eps2};
  const int nln = a.g2 ? 2 : 1;
  for (int l = 0; l < nln; ++l) {
    if (l > 0) {
      sum = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = (tid + 256 * j) * 4;
        if (c < a.D) sum += v[j].x + v[j].y + v[j].z + v[j].w;
      }
    }
    const float mean = block_sum_256(sum, red) / (float)a.D;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = (tid + 256 * j) * 4;
      if (c < a.D) {
        const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
        sq += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    const float var = block_sum_256(sq, red) / (float)a.D;
    if (a.guard && l == 0 && tid == 0 && !(var < INFINITY)) atomicAdd(a.guard, 1);  // NaN / inf in the row: an operand overflowed upstream
    const float rstd = rsqrtf(var + epss[l]);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = (tid + 256 * j) * 4;
      if (c < a.D) {
        const float4 g = *(const float4*)(gs[l] + c);
        const float4 b = *(const float4*)(bs[l] + c);
        v[j] = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                           (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
      }
    }
  }
  emit(v);
}

// Narrow rows (D <= 1024): one float4 per thread and every operand the row needs - input, bias, split-K slabs, affine
// parameters - requested before the first use, so the row costs one memory round trip instead of three (input, slabs,
// affine).  NSLAB / BIAS / RMS are compile-time so no branch sits between the requests.  Same arithmetic order as the
// generic kernel: bias, slabs in order, two-pass variance.
template <typename T, int NSLAB, bool BIAS, bool RMS>
__global__ __launch_bounds__(256) void rownorm_narrow_kernel(RowNormArgs a) {
  __shared__ float red[8];  // one 4-float array per reduction: no barrier is needed to recycle it
  const int row = blockIdx.x, tid = threadIdx.x;
  const bool live = tid * 4 < a.D;
  const int c = min(tid * 4, a.D - 4);  // idle lanes re-read the last quad (no branch), masked below
  float* xr = a.x + (size_t)row * a.ldx;
  const float* src = a.x_in ? a.x_in + (size_t)row * a.ldxin : xr;
  float4 t = *(const float4*)(src + c);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (BIAS) bv = *(const float4*)(a.add_bias + c);
  float4 sl[NSLAB > 0 ? NSLAB : 1];
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) sl[s] = *(const float4*)(a.add_slabs + (size_t)s * a.slab_stride + (size_t)row * a.ldslab + c);
  const float4 g = *(const float4*)(a.g1 + c);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!RMS) b = *(const float4*)(a.b1 + c);
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (BIAS) { t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w; }
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) { t.x += sl[s].x; t.y += sl[s].y; t.z += sl[s].z; t.w += sl[s].w; }
  if (a.write_x && live) *(float4*)(xr + c) = t;
  if (!live) t = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 y;
  if constexpr (RMS) {
    const float sq = block_sum_256_fresh(t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w, red);
    if (a.guard && tid == 0 && !(sq < INFINITY)) atomicAdd(a.guard, 1);
    const float nrm = sqrtf(sq) * rsqrtf((float)a.D);
    const float inv = 1.0f / fmaxf(nrm, a.eps1);
    y = make_float4(t.x * inv * g.x, t.y * inv * g.y, t.z * inv * g.z, t.w * inv * g.w);
  } else {
    const float mean = block_sum_256_fresh(t.x + t.y + t.z + t.w, red) / (float)a.D;
    const float dx = t.x - mean, dy = t.y - mean, dz = t.z - mean, dw = t.w - mean;
    const float var = block_sum_256_fresh(live ? dx * dx + dy * dy + dz * dz + dw * dw : 0.f, red + 4) / (float)a.D;
    if (a.guard && tid == 0 && !(var < INFINITY)) atomicAdd(a.guard, 1);  // NaN / inf in the row: an operand overflowed upstream
    const float rstd = rsqrtf(var + a.eps1);
    y = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  }
  if (live) {
    if (a.out_t) *(typename Vec<T>::x4*)((T*)a.out_t + (size_t)row * a.ldot + c) = pack4<T>(y.x, y.y, y.z, y.w);
    if (a.out_f32) *(float4*)(a.out_f32 + (size_t)row * a.ldo32 + c) = y;
  }
}

template <typename T, int NSLAB>
static void rownorm_narrow_dispatch(const ProfScope& ps, const RowNormArgs& a, hipStream_t stream) {
  const bool bias = a.add_bias != nullptr, rms = a.mode == NORM_RMS;
  const int grid = a.M;
  if (bias && rms) launch_timed(ps, rownorm_narrow_kernel<T, NSLAB, true, true>, dim3(grid), dim3(256), 0, stream, a);
  else if (bias) launch_timed(ps, rownorm_narrow_kernel<T, NSLAB, true, false>, dim3(grid), dim3(256), 0, stream, a);
  else if (rms) launch_timed(ps, rownorm_narrow_kernel<T, NSLAB, false, true>, dim3(grid), dim3(256), 0, stream, a);
  else launch_timed(ps, rownorm_narrow_kernel<T, NSLAB, false, false>, dim3(grid), dim3(256), 0, stream, a);
}

template <typename T>
static bool rownorm_narrow_launch(const ProfScope& ps, const RowNormArgs& a, hipStream_t stream) {
  if (a.D > 1024 || a.mode == NORM_NONE || a.g2 != nullptr || a.f32_slot != nullptr) return false;
  switch (a.nslab) {
    case 0: rownorm_narrow_dispatch<T, 0>(ps, a, stream); return true;
    case 1: rownorm_narrow_dispatch<T, 1>(ps, a, stream); return true;
    case 2: rownorm_narrow_dispatch<T, 2>(ps, a, stream); return true;
    case 4: rownorm_narrow_dispatch<T, 4>(ps, a, stream); return true;
    case 8: rownorm_narrow_dispatch<T, 8>(ps, a, stream); return true;
    default: return false;
  }
}

// Wave-per-row variant for D <= 1024: no block barriers, reductions are register shuffles only, four rows per
// 256-thread block.  Same arithmetic order per row as the block variant is NOT required (tests compare to torch).
template <typename T>
__global__ __launch_bounds__(256) void rownorm_wave_kernel(RowNormArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  constexpr int J = 4;
  float4 v[J];
  float* xr = a.x + (size_t)row * a.ldx;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (c < a.D) {
      float4 t = a.x_in ? *(const float4*)(a.x_in + (size_t)row * a.ldxin + c) : *(const float4*)(xr + c);
      if (a.add_bias) {
        const float4 b = *(const float4*)(a.add_bias + c);
        t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
      }
      for (int s = 0; s < a.nslab; ++s) {
        const float4 p = *(const float4*)(a.add_slabs + (size_t)s * a.slab_stride + (size_t)row * a.ldslab + c);
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      if (a.write_x) *(float4*)(xr + c) = t;
      v[j] = t;
      sum += t.x + t.y + t.z + t.w;
    } else {
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (a.mode == NORM_NONE) return;
  if (a.mode == NORM_RMS) {
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) sq += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    sq = wave_sum(sq);
    if (a.guard && lane == 0 && !(sq < INFINITY)) atomicAdd(a.guard, 1);
    const float nrm = sqrtf(sq) * rsqrtf((float)a.D);
    const float inv = 1.0f / fmaxf(nrm, a.eps1);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < a.D) {
        const float4 g = *(const float4*)(a.g1 + c);
        v[j] = make_float4(v[j].x * inv * g.x, v[j].y * inv * g.y, v[j].z * inv * g.z, v[j].w * inv * g.w);
      }
    }
  } else {
    const float* gs[2] = {a.g1, a.g2};
    const float* bs[2] = {a.b1, a.b2};
    const float epss[2] = {a.eps1, a.eps2};
    const int nln = a.g2 ? 2 : 1;
    for (int l = 0; l < nln; ++l) {
      if (l > 0) {
        sum = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int c = (lane + 64 * j) * 4;
          if (c < a.D) sum += v[j].x + v[j].y + v[j].z + v[j].w;
        }
      }
      const float mean = wave_sum(sum) / (float)a.D;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < a.D) {
          const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
          sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
      }
      const float var = wave_sum(sq) / (float)a.D;
      if (a.guard && l == 0 && lane == 0 && !(var < INFINITY)) atomicAdd(a.guard, 1);
      const float rstd = rsqrtf(var + epss[l]);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < a.D) {
          const float4 g = *(const float4*)(gs[l] + c);
          const float4 b = *(const float4*)(bs[l] + c);
          v[j] = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                             (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (c < a.D) {
      if (a.out_t) *(typename Vec<T>::x4*)((T*)a.out_t + (size_t)row * a.ldot + c) = pack4<T>(v[j].x, v[j].y, v[j].z, v[j].w);
      if (a.out_f32) *(float4*)(a.out_f32 + (size_t)row * a.ldo32 + c) = v[j];
    }
  }
}

int rownorm_launch(int dtype, const RowNormArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.M > 0 && a.D > 0 && a.D % 4 == 0 && a.D <= 4096, "rownorm: bad shape M=%d D=%d", a.M, a.D);
  TT_REQUIRE(a.ldx % 4 == 0, "rownorm: ldx must be a multiple of 4");
  ProfScope ps(PROF_ROWNORM, stream, 0.0, (double)a.M * a.D * (4.0 * (1 + a.nslab + (a.write_x ? 1 : 0)) + (a.out_t ? 2.0 : 0.0) + (a.out_f32 ? 4.0 : 0.0)), true);
  // few rows (decode): one block per row keeps 4x more loads in flight; many rows: wave per row, no barriers
  if (a.D <= 1024 && a.M >= 1024 && !a.row_blocks) {
    TT_DISPATCH_T(dtype, T, launch_timed(ps, rownorm_wave_kernel<T>, dim3(cdiv(a.M, 4)), dim3(256), 0, stream, a));
  } else {
    bool narrow = false;
    TT_DISPATCH_T(dtype, T, narrow = rownorm_narrow_launch<T>(ps, a, stream));
    if (!narrow) TT_DISPATCH_T(dtype, T, launch_timed(ps, (rownorm_kernel<T, -1>), dim3(a.M), dim3(256), 0, stream, a));
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------- group norm
// x: [B][S][C] f32 token-major, 32 groups of cpg = C/32 channels (cpg % 4 == 0), C/4 a power of
// two <= 256 or C == 1024*k.  Stage 1 writes per-(batch, row-chunk, group) (sum, sumsq).
constexpr int GN_ROWS = 16;
constexpr int GN_MAX_CHUNKS = 64;  // the apply kernel's finalize prologue reduces <= 8 partials per thread
static inline int gn_rows_per_chunk(int S) { return std::max(GN_ROWS, cdiv(S, GN_MAX_CHUNKS)); }

// (vl: valid rows of sample b in a padded batch, see GroupNormArgs::vlen; a chunk past them contributes zeros)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int S, int C, float* __restrict__ partial, int rows_per_chunk,
                                                       GroupNormArgs va) {
  __shared__ float ls[256][2];
  const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
  const int vl = va.vperiod > 0 ? va.vlen[b % va.vperiod] : S;
  const int tid = threadIdx.x;
  const int c4n = C >> 2;                       // float4 columns per row
  const int CL = c4n < 256 ? c4n : 256;         // column lanes
  const int RL = 256 / CL;                      // row lanes
  const int cl = tid % CL, rl = tid / CL;
  const int cpg4 = (C / 32) >> 2;               // float4 columns per group
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(vl, r0 + rows_per_chunk);
  // With C > 1024 a thread owns several columns in different groups: handle one column set per pass.
  for (int cb = 0; cb < c4n; cb += 256) {
    float s = 0.f, q = 0.f;
    const int c4 = cb + cl;
    if (c4 < c4n) {
#pragma unroll 8
      for (int r = r0 + rl; r < r1; r += RL) {
        const float4 t = *(const float4*)(x + ((size_t)b * S + r) * C + c4 * 4);
        s += t.x + t.y + t.z + t.w;
        q += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
      }
    }
    __syncthreads();
    ls[tid][0] = s;
    ls[tid][1] = q;
    __syncthreads();
    // groups covered by this pass: columns [cb, cb+CL) -> groups (cb/cpg4) .. ; one thread per group
    const int ngrp = CL / cpg4;
    if (tid < ngrp) {
      float ts = 0.f, tq = 0.f;
      for (int r = 0; r < RL; ++r)
        for (int c = 0; c < cpg4; ++c) {
          ts += ls[r * CL + tid * cpg4 + c][0];
          tq += ls[r * CL + tid * cpg4 + c][1];
        }
      const int g = cb / cpg4 + tid;
      float* p = partial + (((size_t)b * nchunk + chunk) * 32 + g) * 2;
      p[0] = ts;
      p[1] = tq;
    }
  }
}

// Per-(batch, group) mean / rstd from partial sums, fp64 combine in a fixed order.  Two sources:
//   a.gemm_part == nullptr : partial[b][chunk][32][2] from gn_stats_kernel
//   a.gemm_part != nullptr : [row_tile][slot][C/16][2] written by the producing GEMM's epilogue
// The first GN_HEAD fused partials of every thread can be requested ahead of the activation rows (gn_partial_head) so the
// statistics round trip overlaps the row loads; gn_finalize then sums head + remainder in the same fixed order.
constexpr int GN_HEAD = 8;

// All index arithmetic here is shifts and compares: part_rows (the producing GEMM's wave-tile height) is a power of two
// and the strips per group (C / 32 / 16) are a compile-time power of two on the C == 1024 path (SPG_SHIFT = 1), 
// because this code sits in front of every activation load of the kernel (integer divisions by run-time values cost
// ~30 instructions each, 3 per partial item).
template <int SPG_SHIFT>
__device__ __forceinline__ float2 gn_partial_item(const GroupNormArgs& a, int b, int g, int e, int t0, int nitems, int nc16, int r_shift) {
  const int ec = min(e, nitems - 1);  // clamped, unconditional load; out-of-range items are zeroed by the caller
  const int t = t0 + (ec >> SPG_SHIFT), strip = (g << SPG_SHIFT) + (ec & ((1 << SPG_SHIFT) - 1));
  // a row tile's slot 0 holds the rows of the sequence its FIRST row belongs to, slot 1 those of the next sequence: tile t of
  // sample b starts inside sample b unless it is the first tile and straddles in from sample b - 1
  const int slot = ((t << r_shift) < b * a.S) ? 1 : 0;
  return *(const float2*)(a.gemm_part + (((size_t)t * 2 + slot) * nc16 + strip) * 2);
}

template <int SPG_SHIFT>
__device__ __forceinline__ void gn_partial_head(const GroupNormArgs& a, int b, int tid, float2 (&head)[GN_HEAD]) {
  const int g = tid & 31, part = tid >> 5;
  const int S = a.S, r_shift = 31 - __builtin_clz(a.part_rows), nc16 = a.C >> 4;
  const int t0 = (b * S) >> r_shift, t1 = ((b + 1) * S - 1) >> r_shift;
  const int nitems = (t1 - t0 + 1) << SPG_SHIFT;
#pragma unroll
  for (int k = 0; k < GN_HEAD; ++k) head[k] = gn_partial_item<SPG_SHIFT>(a, b, g, part + 8 * k, t0, nitems, nc16, r_shift);
}

template <int SPG_SHIFT>
__device__ __forceinline__ void gn_finalize(const GroupNormArgs& a, int b, int tid, int nchunk, float* mean_s, float* rstd_s,
                                            double (*part_s)[32], double (*part_q)[32], const float2* head = nullptr) {
  const int g = tid & 31, part = tid >> 5;
  double s = 0.0, q = 0.0;
  if (a.gemm_part) {
    const int S = a.S, r_shift = 31 - __builtin_clz(a.part_rows), nc16 = a.C >> 4;
    const int t0 = (b * S) >> r_shift, t1 = ((b + 1) * S - 1) >> r_shift;
    const int nitems = (t1 - t0 + 1) << SPG_SHIFT;  // 16-column strips per group x row tiles
    int e = part;
    if (head) {
#pragma unroll
      for (int k = 0; k < GN_HEAD; ++k, e += 8) {
        if (e < nitems) {
          s += (double)head[k].x;
          q += (double)head[k].y;
        }
      }
    }
#pragma unroll 4
    for (; e < nitems; e += 8) {
      const float2 v = gn_partial_item<SPG_SHIFT>(a, b, g, e, t0, nitems, nc16, r_shift);
      s += (double)v.x;
      q += (double)v.y;
    }
  } else {
    float2 pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // nchunk <= 64: at most 8 independent loads per thread
      const int i = part + 8 * k;
      pv[k] = i < nchunk ? *(const float2*)(a.partial + (((size_t)b * nchunk + i) * 32 + g) * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += (double)pv[k].x;
      q += (double)pv[k].y;
    }
  }
  part_s[part][g] = s;
  part_q[part][g] = q;
  __syncthreads();
  if (tid < 32) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      ss += part_s[p][tid];
      qq += part_q[p][tid];
    }
    // the sums are combined in fp64 (E[x^2] - E[x]^2 cancels in fp32); the reciprocal square root of the O(1) result is an
    // fp32 instruction, not an fp64 divide + square root (hundreds of cycles on the one wave every block waits for)
    // 1 / (S * C / 32), set by groupnorm_launch; a padded batch counts the sample's valid rows only
    const double inv_n = a.vperiod > 0 ? 1.0 / ((double)a.vlen[b % a.vperiod] * (double)(a.C / 32)) : a.inv_count;
    const double m = ss * inv_n;
    double var = qq * inv_n - m * m;
    if (a.guard && !(var < 1.0e300)) atomicAdd(a.guard, 1);  // NaN / inf statistics: an operand overflowed upstream
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)m;
    rstd_s[tid] = rsqrtf((float)var + a.eps);
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GroupNormArgs a, int nchunk, int rows_per_chunk, int rows_per_block) {
  __shared__ float mean_s[32], rstd_s[32];
  __shared__ double part_s[8][32], part_q[8][32];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int C = a.C, S = a.S;
  {
    const int spg = (a.C / 32) >> 4;  // strips per group: 1, 2 or 4 (C = 512, 1024, 2048)
    if (spg == 4) gn_finalize<2>(a, b, tid, nchunk, mean_s, rstd_s, part_s, part_q);
    else if (spg == 2) gn_finalize<1>(a, b, tid, nchunk, mean_s, rstd_s, part_s, part_q);
    else gn_finalize<0>(a, b, tid, nchunk, mean_s, rstd_s, part_s, part_q);
  }
  const int c4n = C >> 2;
  const int cpg = C / 32;
  const int r0 = chunk * rows_per_block;
  const int r1 = min(S, r0 + rows_per_block);
  const int total = (r1 - r0) * c4n;
  const int vl = a.vperiod > 0 ? a.vlen[b % a.vperiod] : S;
  for (int f = tid; f < total; f += 256) {
    const int r = r0 + f / c4n;
    const int c = (f % c4n) * 4;
    const int g = c / cpg;
    const size_t off = ((size_t)b * S + r);
    const float4 t = *(const float4*)(a.x + off * C + c);
    const float4 gm = *(const float4*)(a.gamma + c);
    const float4 bt = *(const float4*)(a.beta + c);
    const float mu = mean_s[g], rs = rstd_s[g];
    float y[4] = {(t.x - mu) * rs * gm.x + bt.x, (t.y - mu) * rs * gm.y + bt.y, (t.z - mu) * rs * gm.z + bt.z,
                  (t.w - mu) * rs * gm.w + bt.w};
    if (a.scale_shift) {
      const float* ss = a.scale_shift + (size_t)(b / (a.ss_batch_div > 0 ? a.ss_batch_div : 1)) * a.ss_batch_stride;
      const float4 sc = *(const float4*)(ss + c);
      const float4 sh = *(const float4*)(ss + C + c);
      y[0] = y[0] * (1.f + sc.x) + sh.x;
      y[1] = y[1] * (1.f + sc.y) + sh.y;
      y[2] = y[2] * (1.f + sc.z) + sh.z;
      y[3] = y[3] * (1.f + sc.w) + sh.w;
    }
    if (a.act != ACT_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = apply_act(y[i], a.act, 0.f);
    }
    if (r >= vl) y[0] = y[1] = y[2] = y[3] = 0.f;  // padding rows of a shorter sequence: exact zeros (the next conv's zero padding)
    if (a.out_t) *(typename Vec<T>::x4*)((T*)a.out_t + off * a.ldot + c) = pack4<T>(y[0], y[1], y[2], y[3]);
    if (a.out_f32) *(float4*)(a.out_f32 + off * a.ldo32 + c) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

// C == 1024 fast path: thread t owns channels 4t..4t+3 (group t/8) of GN_APPLY_ROWS consecutive rows.  The x rows
// are requested BEFORE the statistics are finalised, so the streaming loads overlap the (latency-bound) prologue.
constexpr int GN_APPLY_ROWS = 4;  // 4 rows per block: 2 blocks per CU at the denoiser's 1740 rows, so one block's statistics prologue overlaps the other's stream
// ROWS per block: 4 in general (two blocks per CU overlap each other's statistics prologue); 2 for passes of <= 4096 rows (the
// denoiser alone, 1740 rows: 870 blocks instead of 435 - in-situ A/B -1.7 % on the sampler iteration; 1 row and 8 rows are slower)
template <typename T, bool FUSED, bool SS, int ROWS>
__global__ __launch_bounds__(256) void gn_apply_c1024_kernel(GroupNormArgs a, int nchunk) {
  __shared__ float mean_s[32], rstd_s[32];
  __shared__ double part_s[8][32], part_q[8][32];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int S = a.S;
  constexpr int C = 1024;
  const int r0 = blockIdx.x * ROWS;
  const int c = tid * 4;
  // request order = need order: statistics partials, then the rows and the affine parameters
  float2 head[GN_HEAD];
  if constexpr (FUSED) gn_partial_head<1>(a, b, tid, head);  // FUSED <=> a.gemm_part != nullptr (compile time: no branch to sink consumers into)
  float4 xr[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int r = min(r0 + i, S - 1);
    xr[i] = *(const float4*)(a.x + ((size_t)b * S + r) * C + c);
  }
  const float4 gm = *(const float4*)(a.gamma + c);
  const float4 bt = *(const float4*)(a.beta + c);
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
  if constexpr (SS) {  // SS <=> a.scale_shift != nullptr
    const float* ss = a.scale_shift + (size_t)(b / (a.ss_batch_div > 0 ? a.ss_batch_div : 1)) * a.ss_batch_stride;
    sc = *(const float4*)(ss + c);
    sh = *(const float4*)(ss + C + c);
  }
  __builtin_amdgcn_sched_barrier(0);  // keep every request above in flight before the first consumer waits
  gn_finalize<1>(a, b, tid, nchunk, mean_s, rstd_s, part_s, part_q, FUSED ? head : nullptr);
  const float mu = mean_s[tid >> 3], rs = rstd_s[tid >> 3];
  const int vl = a.vperiod > 0 ? a.vlen[b % a.vperiod] : S;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int r = r0 + i;
    if (r >= S) break;
    const float4 t = xr[i];
    float y[4] = {(t.x - mu) * rs * gm.x + bt.x, (t.y - mu) * rs * gm.y + bt.y, (t.z - mu) * rs * gm.z + bt.z,
                  (t.w - mu) * rs * gm.w + bt.w};
    if constexpr (SS) {
      y[0] = y[0] * (1.f + sc.x) + sh.x;
      y[1] = y[1] * (1.f + sc.y) + sh.y;
      y[2] = y[2] * (1.f + sc.z) + sh.z;
      y[3] = y[3] * (1.f + sc.w) + sh.w;
    }
    if (a.act != ACT_NONE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = apply_act(y[k], a.act, 0.f);
    }
    if (r >= vl) y[0] = y[1] = y[2] = y[3] = 0.f;  // padding rows of a shorter sequence: exact zeros
    const size_t off = (size_t)b * S + r;
    if (a.out_t) *(typename Vec<T>::x4*)((T*)a.out_t + off * a.ldot + c) = pack4<T>(y[0], y[1], y[2], y[3]);
    if (a.out_f32) *(float4*)(a.out_f32 + off * a.ldo32 + c) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

int groupnorm_launch(int dtype, const GroupNormArgs& a0, hipStream_t stream) {
  GroupNormArgs a = a0;
  a.inv_count = 1.0 / ((double)a.S * (double)(a.C / 32));
  const int c4n = a.C / 4;
  TT_REQUIRE(a.C % 128 == 0 && ((c4n <= 256 && (c4n & (c4n - 1)) == 0) || c4n % 256 == 0), "groupnorm: unsupported C=%d", a.C);
  TT_REQUIRE(a.B > 0 && a.S > 0 && a.partial != nullptr, "groupnorm: bad arguments");
  const int rpc = gn_rows_per_chunk(a.S);
  const int nchunk = cdiv(a.S, rpc);
  dim3 grid(nchunk, a.B);
  // (dispatch-timed when the statistics come from the producing GEMM: then the apply kernel is the only launch of this scope)
  ProfScope ps(PROF_GROUPNORM, stream, 0.0, (double)a.B * a.S * a.C * ((a.gemm_part ? 4.0 : 8.0) + (a.out_t ? 2.0 : 0.0) + (a.out_f32 ? 4.0 : 0.0)), a.gemm_part != nullptr);
  if (a.gemm_part) {
    const int spg = (a.C / 32) / 16;
    TT_REQUIRE((a.C / 32) % 16 == 0 && (spg == 1 || spg == 2 || spg == 4) && a.part_rows > 0 && (a.part_rows & (a.part_rows - 1)) == 0 && a.S >= a.part_rows,
               "groupnorm: fused statistics need 16 / 32 / 64 channels per group, a power-of-two row tile and S >= the row tile");
  } else {
    gn_stats_kernel<<<grid, 256, 0, stream>>>(a.x, a.S, a.C, a.partial, rpc, a);
    TT_CHECK_HIP(hipGetLastError());
  }
  const bool few = a.C == 1024 && (long)a.B * a.S <= 4096;
  const int rpb = few ? 2 : GN_APPLY_ROWS;  // apply is pure streaming: many small blocks
  dim3 grid2(cdiv(a.S, rpb), a.B);
  if (a.C == 1024) {
    const int variant = (dtype == DT_BF16 ? 0 : dtype == DT_F16 ? 4 : 8) + (a.gemm_part ? 2 : 0) + (a.scale_shift ? 1 : 0);
#define TT_GN(T, F, SSV)                                                                                              \
    do {                                                                                                                \
      if (few) launch_timed(ps, gn_apply_c1024_kernel<T, F, SSV, 2>, grid2, dim3(256), 0, stream, a, nchunk);          \
      else launch_timed(ps, gn_apply_c1024_kernel<T, F, SSV, GN_APPLY_ROWS>, grid2, dim3(256), 0, stream, a, nchunk);  \
    } while (0)
    switch (variant) {
      case 0: TT_GN(bf16, false, false); break;
      case 1: TT_GN(bf16, false, true); break;
      case 2: TT_GN(bf16, true, false); break;
      case 3: TT_GN(bf16, true, true); break;
      case 4: TT_GN(f16, false, false); break;
      case 5: TT_GN(f16, false, true); break;
      case 6: TT_GN(f16, true, false); break;
      case 7: TT_GN(f16, true, true); break;
      case 8: TT_GN(float, false, false); break;   // (the fp32 verification mode)
      case 9: TT_GN(float, false, true); break;
      case 10: TT_GN(float, true, false); break;
      default: TT_GN(float, true, true); break;
    }
#undef TT_GN
  } else {
    TT_DISPATCH_T(dtype, T, launch_timed(ps, gn_apply_kernel<T>, grid2, dim3(256), 0, stream, a, nchunk, rpc, rpb));
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t groupnorm_partial_floats(int B, int S) { return (size_t)B * cdiv(S, gn_rows_per_chunk(S)) * 32 * 2; }

}  // namespace tt
