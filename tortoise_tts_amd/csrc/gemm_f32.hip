// fp32-operand GEMM of the VERIFICATION mode (DT_F32; tests only - include/tortoise_mi355x.h TT_F32).  Same contract as the product
// GEMM family (gemm.h: conv taps / dilation, second activation source, split-K slabs, every epilogue incl. GroupNorm statistics and
// the QKV scatters), same 64 x 64 tile with 32-row statistics strips, but operands stay fp32 end to end: v_mfma_f32_16x16x4_f32 on
// tiles staged through the LDS with plain loads.  Nothing here is tuned - it exists so that the engines can be held against the
// reference's fp32 modules at fp32 tolerances instead of inside bf16 / fp16 operand noise.
#include "gemm_impl.h"

namespace tt {

template <typename Epi, bool CONV>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDev<typename Epi::Args> g) {
  constexpr int BM = 64, BN = 64, BK = 64, LD = BK + 4;
  __shared__ float As[BM][LD];
  __shared__ float Ws[BN][LD];
  const GemmCore& c = g.c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int bx = (int)(blockIdx.x % c.gx), by = (int)(blockIdx.x / c.gx);
  const int m0 = bx * BM, n0 = by * BN;
  const int z = blockIdx.z;
  const int kt_begin = z * c.sk_quot + min(z, c.sk_rem);
  const int kt_end = kt_begin + c.sk_quot + (z < c.sk_rem ? 1 : 0);
  const float* A = (const float*)c.A;
  const float* W = (const float*)c.W;
  const float* A2 = nullptr;
  if (!CONV && c.A2) A2 = (const float*)c.A2 + (c.a2_slot ? (size_t)(*c.a2_slot) * c.a2_slot_stride : 0);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  typename Epi::template Ops<2, 2> eo;
  Epi::template fetch<2, 2, false>(c, g.e, eo, m0 + wm * 32, n0 + wn * 32, lane);

  // staging geometry: thread -> (row = tid / 4 [+ 0 .. 63 in one pass], 16 consecutive k of the 64-wide k-tile)
  const int lrow = tid >> 2, lk = (tid & 3) * 16;
  const int fr = lane & 15, fk = lane >> 4;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    int tap = 0, kin = kt * BK;
    if (CONV) {
      tap = kt / c.cin_tiles;
      kin = (kt - tap * c.cin_tiles) * BK;
    }
    {  // A tile: row m0 + lrow
      const int m = m0 + lrow;
      const float* src = nullptr;
      if (m < c.M) {
        if (CONV) {
          unsigned s_;
          const unsigned b = fdiv((unsigned)m, c.seq, s_);
          const int s2 = (int)s_ + (tap - c.taps_half) * c.dil;
          if (s2 >= 0 && s2 < c.seq_len) src = A + ((size_t)b * c.seq_len + s2) * c.lda + kin + lk;
        } else if (A2 != nullptr && kt >= c.a2_tile) {
          src = A2 + (size_t)m * c.lda2 + (kin - c.a2_tile * BK) + lk;
        } else {
          src = A + (size_t)m * c.lda + kin + lk;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = src ? *(const float4*)(src + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&As[lrow][lk + q * 4] = v;
      }
    }
    {  // W tile: row n0 + lrow
      const int n = n0 + lrow;
      const float* src = n < c.N ? W + (size_t)n * c.ldw + (size_t)kt * BK + lk : nullptr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = src ? *(const float4*)(src + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&Ws[lrow][lk + q * 4] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      float fa[2], fw[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fa[j] = As[wm * 32 + j * 16 + fr][k4 * 4 + fk];
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[i] = Ws[wn * 32 + i * 16 + fr][k4 * 4 + fk];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i], fa[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  run_epilogue<Epi, 2, 2, 32, 32, false>(c, g.e, acc, eo, eo.step(), m0 + wm * 32, n0 + wn * 32, lane, z);
}

template <>
int gemm_launch_typed<float>(int epi, const GemmArgs& a, const GemmPlan& plan, hipStream_t stream) {
  TT_REQUIRE(plan.tile == TILE_64x64 && !plan.conv3s, "gemm (fp32 verification mode): planned for the 64 x 64 tile only");
  TT_REQUIRE(a.serial_k <= 1, "gemm (fp32 verification mode): serial split-K is not part of it");
  TT_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0 && (((size_t)a.A | (size_t)a.W) & 15) == 0, "gemm (fp32 verification mode): operands must be 16-byte aligned rows");
  const dim3 grid(plan.core.gx * plan.core.gy, 1, plan.splitk);
  ProfScope ps(plan.prof_id, stream, plan.flops, plan.bytes * 2.0, true);
  const bool conv = a.taps > 1;
  if (epi == EPI_STD) {
    typedef EpiStd<float, -1, -1, -1> E;
    GemmDev<EpiStdArgs> d;
    d.c = plan.core;
    d.e = make_epi_std(a);
    if (conv) launch_timed(ps, gemm_f32_kernel<E, true>, grid, dim3(256), 0, stream, d);
    else launch_timed(ps, gemm_f32_kernel<E, false>, grid, dim3(256), 0, stream, d);
  } else if (epi == EPI_QKV_HEADS) {
    GemmDev<EpiQkvHeadsArgs> d;
    d.c = plan.core;
    memset(&d.e, 0, sizeof(d.e));
    d.e.bias = a.bias; d.e.q = a.q; d.e.k = a.k; d.e.v = a.v; d.e.vt = a.vt; d.e.heads = a.heads; d.e.seq_pad = a.seq_pad; d.e.q_scale = a.q_scale;
    d.e.dmodel = make_fastdiv(a.dmodel);
    launch_timed(ps, gemm_f32_kernel<EpiQkvHeads<float>, false>, grid, dim3(256), 0, stream, d);
  } else if (epi == EPI_QKV_DECODE) {
    GemmDev<EpiQkvDecodeArgs> d;
    d.c = plan.core;
    memset(&d.e, 0, sizeof(d.e));
    d.e.bias = a.bias; d.e.step = a.step; d.e.qbuf = a.qbuf; d.e.kc = a.kc; d.e.vc = a.vc; d.e.heads = a.heads; d.e.tmax = a.tmax; d.e.dmodel_i = a.dmodel;
    d.e.q_scale = a.q_scale;
    d.e.dmodel = make_fastdiv(a.dmodel);
    launch_timed(ps, gemm_f32_kernel<EpiQkvDecode<float>, false>, grid, dim3(256), 0, stream, d);
  } else if (epi == EPI_GEGLU) {
    GemmDev<EpiGegluArgs> d;
    d.c = plan.core;
    d.e.bias = a.bias; d.e.out_t = a.out_t; d.e.ldot = a.ldot;
    launch_timed(ps, gemm_f32_kernel<EpiGeglu<float>, false>, grid, dim3(256), 0, stream, d);
  } else {
    set_error("gemm: unknown epilogue %d", epi);
    return -1;
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

template <>
int gemm_init_typed<float>() { return 0; }  // (static LDS only: nothing to configure)

}  // namespace tt
