// Shared device/host helpers for the MI355X (gfx950) Tortoise engine.
// Everything here targets CDNA4 directly: 64-lane waves, v_mfma_f32_16x16x32_{bf16,f16}.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

namespace tt {

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <typename T> struct Vec;
template <> struct Vec<bf16> {
  typedef __attribute__((ext_vector_type(8))) __bf16 x8;
  typedef __attribute__((ext_vector_type(4))) __bf16 x4;
  typedef __attribute__((ext_vector_type(2))) __bf16 x2;
};
template <> struct Vec<f16> {
  typedef __attribute__((ext_vector_type(8))) _Float16 x8;
  typedef __attribute__((ext_vector_type(4))) _Float16 x4;
  typedef __attribute__((ext_vector_type(2))) _Float16 x2;
};

// float "operands": the verification mode (DT_F32) - every GEMM / attention operand stays fp32, so the engine can be held against the
// reference's fp32 modules at fp32 tolerances (slow kernels, tests only)
template <> struct Vec<float> {
  typedef __attribute__((ext_vector_type(8))) float x8;
  typedef __attribute__((ext_vector_type(4))) float x4;
  typedef __attribute__((ext_vector_type(2))) float x2;
};

// D(16x16 f32) += A(16x32) * B(32x16).  Lane l holds A[row = l&15][k = (l>>4)*8 .. +7],
// B[k = (l>>4)*8 .. +7][col = l&15]; D lane l reg r = D[row = (l>>4)*4 + r][col = l&15].
__device__ __forceinline__ f32x4 mfma16(Vec<bf16>::x8 a, Vec<bf16>::x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(Vec<f16>::x8 a, Vec<f16>::x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// D(32x32 f32) += A(32x16) * B(16x32).  Lane l holds A[row = l&31][k = (l>>5)*8 .. +7], B[k = (l>>5)*8 .. +7][col = l&31];
// D lane l reg r = D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(Vec<bf16>::x8 a, Vec<bf16>::x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(Vec<f16>::x8 a, Vec<f16>::x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <typename T> __device__ __forceinline__ typename Vec<T>::x4 pack4(float a, float b, float c, float d) {
  typename Vec<T>::x4 r;
  r[0] = (T)a; r[1] = (T)b; r[2] = (T)c; r[3] = (T)d;
  return r;
}

// Cross-lane exchanges without the LDS: __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt (a ~100-cycle LDS round trip per
// step, 6 of them in a wave reduction, and at one or two waves per SIMD nothing hides them).  Within a row of 16 lanes the
// DPP modifier does the exchange inside the VALU instruction; across rows gfx950 has v_permlane16_swap / v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {  // CTRL: quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_mirror 0x140, row_half_mirror 0x141
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// (value of lane l, value of lane l ^ 16) / (l, l ^ 32) as a pair, in lane-symmetric order.  Inline asm: with ROCm 7.2's
// __builtin_amdgcn_permlane16_swap / permlane32_swap both elements of the returned pair come out as the FIRST register
// (`v_add_f32 v1, v1, v1` after the swap), i.e. the builtin silently drops the second result.
__device__ __forceinline__ void pair_xor16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void pair_xor32(float v, float& a, float& b) {
  a = v;
  b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float add_xor8(float v) { return v + dpp_mov<0x128>(v); }   // row_ror:8 == lane ^ 8 inside a 16-lane row
__device__ __forceinline__ float add_xor16(float v) { float a, b; pair_xor16(v, a, b); return a + b; }
__device__ __forceinline__ float add_xor32(float v) { float a, b; pair_xor32(v, a, b); return a + b; }
__device__ __forceinline__ float max_xor16(float v) { float a, b; pair_xor16(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float max_xor32(float v) { float a, b; pair_xor32(v, a, b); return fmaxf(a, b); }

__device__ __forceinline__ float wave_sum(float v) {  // every lane gets the sum of all 64
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the 8
  v += dpp_mov<0x140>(v);  // row_mirror: the other half of the row
  v = add_xor16(v);
  return add_xor32(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  v = max_xor16(v);
  return max_xor32(v);
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
// same without the leading barrier: `red` must not have been read since the last barrier (give every reduction of a kernel its own array)
__device__ __forceinline__ float block_sum_256_fresh(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float gelu_tanh(float x) {  // HF "gelu_new": 0.5 x (1 + tanh(u)) == x * sigmoid(2u), one v_exp + one v_rcp
  const float k2 = 2.0f * 0.7978845608028654f;
  const float u2 = k2 * (x + 0.044715f * x * x * x);
  return x * __frcp_rn(1.0f + __expf(-u2));
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

enum Act { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_LRELU = 4, ACT_TANH = 5 };
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_GELU_TANH: return gelu_tanh(v);
    case ACT_GELU_ERF: return gelu_erf(v);
    case ACT_SILU: return silu(v);
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);
const char* last_error();

#define TT_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      tt::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                        \
    }                                                                                   \
  } while (0)

#define TT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      tt::set_error(__VA_ARGS__);        \
      return -1;                         \
    }                                    \
  } while (0)

#define TT_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

enum DType { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };
static inline int dtype_bytes(int dtype) { return dtype == DT_F32 ? 4 : 2; }
// f(tag) with tag.v of the operand type: TT_DISPATCH_T(dtype, T, stmt) runs stmt with T = bf16 / f16 / float
#define TT_DISPATCH_T(dtype, T, ...)                                   \
  do {                                                                 \
    if ((dtype) == tt::DT_BF16) { typedef tt::bf16 T; __VA_ARGS__; }   \
    else if ((dtype) == tt::DT_F16) { typedef tt::f16 T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }                             \
  } while (0)

}  // namespace tt

// ---------------------------------------------------------------- per-kernel-class timing (bench.py roofline leg)
// When enabled (tt_prof_enable), every launcher brackets its kernel with HIP events recorded on the
// launch stream and accumulates the algorithmic flops / bytes of the launch.  Not used on the product
// path (graphs stay enabled there); bench.py runs one extra un-captured step with it.
namespace tt {
enum ProfId {  // one class per kernel instantiation that actually runs (names: common.hip)
  PROF_GEMM_64x64_STD = 0, PROF_GEMM_64x64_CONV, PROF_GEMM_64x64_QKV, PROF_GEMM_64x64_QKVDEC,
  PROF_GEMM_128x64_STD, PROF_GEMM_128x64_CONV, PROF_GEMM_128x64_QKV, PROF_GEMM_128x64_QKVDEC,
  PROF_GEMM_128x128_STD, PROF_GEMM_128x128_CONV, PROF_GEMM_128x128_QKV, PROF_GEMM_128x128_QKVDEC,
  PROF_GEMM_256x256_STD, PROF_GEMM_256x256_CONV, PROF_GEMM_256x256_QKV, PROF_GEMM_256x256_QKVDEC,
  PROF_FLASH, PROF_DECODE_ATTN, PROF_ROWNORM, PROF_GROUPNORM, PROF_SAMPLE, PROF_GLUE, PROF_CONV1D, PROF_CONVT, PROF_LVC,
  PROF_GEMM_64x64_STATS,  // the denoiser's 1x1 GEMMs with the GroupNorm-statistics epilogue (M = 1740): kept apart from the decode GEMMs of the same tile
  PROF_GEMM_GNA,          // GEMM with the GroupNorm apply on its A path (gemm_gna.h)
  PROF_GEMM_32x16_STD, PROF_GEMM_32x16_QKVDEC, PROF_GEMM_64x16_STD, PROF_GEMM_64x16_QKVDEC,  // skinny decode tiles (small batches)
  PROF_GEMV,              // GEMV-shaped decode GEMMs of handles with max_batch <= 4 (gemv.hip)
  PROF_COUNT
};
extern bool g_prof_on;
void prof_record(int id, hipStream_t s, bool begin, double flops, double bytes);
void prof_pair(int id, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
// Two timing forms.  Bracket (default): an event recorded on the launch stream before and after the launcher's kernels - right
// for launchers that enqueue several kernels, but the two hipEventRecord packets add ~2.5 us to a 7 us kernel.  Dispatch
// (`dispatch = true`, single-kernel launchers): the scope only hands out an event pair and the launcher passes it to
// hipExtLaunchKernelGGL, which stamps the pair with the start / end timestamps of the dispatch itself - the same clock a
// rocprofv3 kernel trace reads, so the roofline leg of bench.py and profiles/ agree.
struct ProfScope {
  int id; hipStream_t s; bool on; bool dispatch;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(int id_, hipStream_t s_, double flops, double bytes, bool dispatch_ = false) : id(id_), s(s_), on(g_prof_on), dispatch(dispatch_) {
    if (!on) return;
    if (dispatch) prof_pair(id, flops, bytes, &e0, &e1);
    else prof_record(id, s, true, flops, bytes);
  }
  ~ProfScope() {
    if (on && !dispatch) prof_record(id, s, false, 0, 0);
  }
};
const char* prof_name(int id);

// Launch `kernel` on `s`; under the profiler (dispatch-form scope) through hipExtLaunchKernelGGL with the scope's event pair.
template <typename... P, typename... A>
static inline void launch_timed(const ProfScope& ps, void (*kernel)(P...), dim3 grid, dim3 block, size_t smem, hipStream_t s, A&&... args) {
  if (ps.on && ps.dispatch) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)smem, s, ps.e0, ps.e1, 0, static_cast<P>(args)...);
  else kernel<<<grid, block, smem, s>>>(static_cast<P>(args)...);
}
}  // namespace tt
