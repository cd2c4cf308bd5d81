// MFMA GEMM for gfx950 (see gemm.h).  256 threads = 4 waves in a 2x2 grid; each wave owns a
// (BM/2)x(BN/2) sub-tile built from v_mfma_f32_16x16x32 fragments.  The MFMA is issued "swapped"
// (W fragment as the A operand, activation fragment as B) so that a lane ends up holding four
// consecutive output columns n..n+3 of one row m: epilogue stores are 16 B (f32) / 8 B (bf16).
// Tiles move global -> registers -> LDS (double buffered, one barrier per 64-deep k-step); the
// register stage is what lets the conv taps shift rows and zero-fill sequence edges for free.
#include "gemm.h"

namespace tt {

// BK (k-depth of one LDS stage) is a template parameter: 64 for the MFMA-bound shapes, 256 for the decode
// shapes (M <= 256), where a block's k-loop is a chain of memory round trips and fewer, fatter stages win.
// LDS row pitch = BK + 8 elements (keeps 16-B alignment, staggers banks).

template <typename T>
struct EpiStd {
  static constexpr int kId = 0;
  // Every epilogue is split in two so that run_epilogue can do ALL arithmetic first and issue ALL stores last:
  // apply() is register-only; bv / rv are the bias and residual quads for (m, n..n+3) fetched ahead (zeros if absent).
  __device__ __forceinline__ void apply(const GemmArgs& g, f32x4& v, const float4& bv, const float4& rv) const {
    if (g.splitk > 1) return;  // raw partial sums; bias / activation / residual belong to the slab consumer
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    if (g.act != ACT_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], g.act, g.slope);
    }
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  }
  __device__ __forceinline__ void store(const GemmArgs& g, int m, int n, const f32x4& v, int nvalid, int z) const {
    if (g.splitk > 1) {
      float* o = g.out_f32 + (size_t)z * g.M * g.ldo32 + (size_t)m * g.ldo32 + n;
      if (nvalid == 4 && (g.ldo32 & 3) == 0) {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
      return;
    }
    if (g.out_f32) {
      float* o = g.out_f32 + (size_t)m * g.ldo32 + n;
      if (nvalid == 4 && (g.ldo32 & 3) == 0) {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
    }
    if (g.out_t) {
      T* o = (T*)g.out_t + (size_t)m * g.ldot + n;
      if (nvalid == 4 && (g.ldot & 3) == 0) {
        *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = (T)v[i];
      }
    }
  }
};

template <typename T>
struct EpiQkvHeads {
  static constexpr int kId = 1;
  __device__ __forceinline__ void apply(const GemmArgs& g, f32x4& v, const float4& bv, const float4&) const {
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  __device__ __forceinline__ void store(const GemmArgs& g, int m, int n, const f32x4& v, int nvalid, int z) const {
    // N == 3 * dmodel and dmodel % 64 == 0, so nvalid is always 4 here.
    const int part = n / g.dmodel;
    const int c = n - part * g.dmodel;
    const int h = c >> 6, d = c & 63;
    const int b = m / g.seq_len, s = m - b * g.seq_len;
    const size_t bh = (size_t)b * g.heads + h;
    if (part == 0) {
      T* o = (T*)g.q + (bh * g.seq_len + s) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0] * g.q_scale, v[1] * g.q_scale, v[2] * g.q_scale, v[3] * g.q_scale);
    } else if (part == 1) {
      T* o = (T*)g.k + (bh * g.seq_len + s) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    } else {
      if (g.v) {
        T* o = (T*)g.v + (bh * g.seq_len + s) * 64 + d;
        *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
      }
      if (g.vt) {
        T* o = (T*)g.vt + (bh * 64 + d) * g.seq_pad + s;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[(size_t)i * g.seq_pad] = (T)v[i];
      }
    }
  }
};

template <typename T>
struct EpiQkvDecode {
  static constexpr int kId = 2;
  __device__ __forceinline__ void apply(const GemmArgs& g, f32x4& v, const float4& bv, const float4&) const {
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  __device__ __forceinline__ void store(const GemmArgs& g, int m, int n, const f32x4& v, int nvalid, int z) const {
    const int part = n / g.dmodel;
    const int c = n - part * g.dmodel;
    const int h = c >> 6, d = c & 63;
    const int t = *g.step;
    const size_t bh = (size_t)m * g.heads + h;
    if (part == 0) {
      T* o = (T*)g.qbuf + (size_t)m * g.dmodel + c;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0] * g.q_scale, v[1] * g.q_scale, v[2] * g.q_scale, v[3] * g.q_scale);
    } else if (part == 1) {
      T* o = (T*)g.kc + ((bh * 8 + (d >> 3)) * g.tmax + t) * 8 + (d & 7);
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    } else {
      T* o = (T*)g.vc + (bh * g.tmax + t) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    }
  }
};

// Epilogue shared by both GEMM kernels.  With g.gn_part set (EPI_STD, f32 output feeding a GroupNorm32) every
// wave also emits (sum, sum of squares) of the values it just produced, per 16-column strip of its TM-row
// tile: gn_part[row_tile][slot][n / 16][2], slot 1 = rows that belong to the NEXT sequence when the row tile
// straddles a sequence boundary.  The GroupNorm apply kernel adds these up in a fixed order (deterministic),
// which removes the separate statistics pass over the tensor.
__device__ __forceinline__ float4 load_upto4(const float* p, int nvalid) {  // ragged / unaligned edge: element loads
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

template <typename Epi, int FM, int FN, int TM, int TN>
__device__ __forceinline__ void run_epilogue(const GemmArgs& g, f32x4 (&acc)[FN][FM], int m0w, int n0w, int lane, int z) {
  const int fr = lane & 15, fg = lane >> 4;
  Epi epi;
  const bool stats = Epi::kId == 0 && g.gn_part != nullptr && g.splitk == 1;
  const int rt = m0w / TM;                              // row-tile index (m0w is a multiple of TM)
  const int b_first = stats ? m0w / g.gn_seq : 0;
  // Bias and residual operands are requested as whole quads for a full column strip BEFORE any arithmetic or store:
  // one memory round trip per strip instead of one per element (a per-element `if (i < nvalid) v += bias[n + i]`
  // compiles to load / s_waitcnt vmcnt(0) / branch chains).  Out-of-range rows / columns are clamped, never stored.
  const bool use_bias = g.bias != nullptr && !(Epi::kId == 0 && g.splitk > 1);
  const bool use_res = Epi::kId == 0 && g.res != nullptr && g.splitk == 1;
  const bool quads = (g.N & 3) == 0 && g.N >= 4 && (!use_bias || ((size_t)g.bias & 15) == 0) &&
                     (!use_res || (((size_t)g.res & 15) == 0 && (g.ldres & 3) == 0));
  // phase 1: every bias / residual quad of the wave tile is requested up front
  float4 bv[FN], rv[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0w + i * 16 + fg * 4;
    const int nvalid = g.N - n >= 4 ? 4 : g.N - n;
    bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < FM; ++j) rv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quads) {
      const int nc = min(n, g.N - 4);
      if (use_bias) bv[i] = *(const float4*)(g.bias + nc);
      if (use_res) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int mc = min(m0w + j * 16 + fr, g.M - 1);
          rv[i][j] = *(const float4*)(g.res + (size_t)mc * g.ldres + nc);
        }
      }
    } else {
      if (use_bias) bv[i] = load_upto4(g.bias + n, nvalid);
      if (use_res) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int m = m0w + j * 16 + fr;
          if (m < g.M) rv[i][j] = load_upto4(g.res + (size_t)m * g.ldres + n, nvalid);
        }
      }
    }
  }
  // phase 2: arithmetic and GroupNorm partial statistics, registers only
  float s0[FN], q0[FN], s1[FN], q1[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0w + i * 16 + fg * 4;
    const int nvalid = g.N - n >= 4 ? 4 : g.N - n;
    s0[i] = q0[i] = s1[i] = q1[i] = 0.f;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0w + j * 16 + fr;
      epi.apply(g, acc[i][j], bv[i], rv[i][j]);
      if (stats) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (r < nvalid && m < g.M) ? acc[i][j][r] : 0.f;
          sv += e;
          qv += e * e;
        }
        const bool first = m / g.gn_seq == b_first;
        s0[i] += first ? sv : 0.f;
        q0[i] += first ? qv : 0.f;
        s1[i] += first ? 0.f : sv;
        q1[i] += first ? 0.f : qv;
      }
    }
  }
  // phase 3: stores, back to back
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0w + i * 16 + fg * 4;
    const int nvalid = g.N - n >= 4 ? 4 : g.N - n;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0w + j * 16 + fr;
      if (m < g.M && n < g.N) epi.store(g, m, n, acc[i][j], nvalid, z);
    }
  }
  if (stats) {
    const bool straddle = (m0w + TM - 1) / g.gn_seq != b_first;  // wave-uniform
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n16 = (n0w + i * 16) >> 4;
      if (n0w + i * 16 < g.N) {
        const float a0 = wave_sum(s0[i]), b0 = wave_sum(q0[i]);
        float a1 = 0.f, b1 = 0.f;
        if (straddle) {
          a1 = wave_sum(s1[i]);
          b1 = wave_sum(q1[i]);
        }
        if (lane == 0) {
          float* p = g.gn_part + (((size_t)rt * 2 + 0) * g.gn_ncol16 + n16) * 2;
          *(float2*)p = make_float2(a0, b0);
          float* p1 = g.gn_part + (((size_t)rt * 2 + 1) * g.gn_ncol16 + n16) * 2;
          *(float2*)p1 = make_float2(a1, b1);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int BK, int NW, typename Epi, bool CONV>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(GemmArgs g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BKP = BK + 8;
  constexpr int NT = NW * 64;               // threads per workgroup
  constexpr int WGN = NW / 2;               // wave grid: 2 (rows) x WGN (columns)
  constexpr int TM = BM / 2, TN = BN / WGN;  // wave tile
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int TPR = BK / 8;        // threads per tile row (16 B each)
  constexpr int RPP = NT / TPR;      // rows per workgroup-wide pass
  constexpr int PA = BM / RPP, PW = BN / RPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As = (T*)smem_raw;                 // [2][BM][BKP]
  T* Ws = As + 2 * BM * BKP;            // [2][BN][BKP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // XCD-aware tile order: hardware dispatches workgroup i to XCD i % 8, each with a private 4 MiB L2.
  // Remap so every XCD owns a contiguous run of tiles (m fastest): its blocks then share W panels and
  // re-use A rows out of ITS L2 instead of all eight L2s each streaming every panel.
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_mode != 0) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nwg = gx * gy;
    const int id = blockIdx.x + gx * blockIdx.y;
    const int xcd = id & 7, loc = id >> 3;
    if (g.xcd_mode == 3 && (gx & 1) == 0 && (gy & 3) == 0) {
      // 2-D ownership: XCD (xm, xn) owns half of the row tiles and a quarter of the column tiles
      const int hx = gx >> 1, qy = gy >> 2;
      const int xm = xcd & 1, xn = xcd >> 1;
      bx = xm * hx + loc % hx;
      by = xn * qy + loc / hx;
    } else {
      const int q = nwg >> 3, r = nwg & 7;
      const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
      if (g.xcd_mode == 2) {  // column tiles fastest: an XCD owns a band of rows and sees every W panel
        by = nid % gy;
        bx = nid / gy;
      } else {                // row tiles fastest: an XCD owns a few W panels and sees every A row
        bx = nid % gx;
        by = nid / gx;
      }
    }
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int z = blockIdx.z;
  const int nk_total = g.K / BK;
  const int kt_begin = (int)((long long)nk_total * z / g.splitk);
  const int kt_end = (int)((long long)nk_total * (z + 1) / g.splitk);

  const int lrow = tid / TPR;
  const int lcol = (tid % TPR) * 8;  // element offset inside the k-tile
  const T* A = (const T*)g.A;
  const T* W = (const T*)g.W;

  // per-pass source rows (conv taps shift them per k-tile)
  int a_b[PA], a_s[PA];
  bool a_ok[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int m = m0 + lrow + RPP * p;
    a_ok[p] = m < g.M;
    if (CONV) {
      a_b[p] = m / g.seq_len;
      a_s[p] = m - a_b[p] * g.seq_len;
    } else {
      a_b[p] = 0;
      a_s[p] = m;
    }
  }

  // Two register tile sets: loads run TWO k-tiles ahead of the MFMAs (one tile being written to LDS,
  // one still in flight), because at these shapes a block's k-step is shorter than the L2/HBM latency.
  x8 ra0[PA], rw0[PW], ra1[PA], rw1[PW];
  unsigned zm0 = 0u, zm1 = 0u;  // CONV only: bit p set = A row p of that set is conv padding (must read as zero)
  const x8 zero8 = {};

  // Loads are unconditional and their results are NOT touched until store_tile: rows beyond M / N are
  // clamped to a valid address and simply produce output rows / columns that the epilogue never stores,
  // so no select is needed (a select right after the load would force an immediate vmcnt wait and
  // destroy the prefetch distance).  Only conv padding needs real zeros; that select happens at store time.
  auto load_tile = [&](x8 (&ra)[PA], x8 (&rw)[PW], unsigned& zm, int kt) {
    const int k0 = kt * BK;
    int tap = 0, kin = k0;
    if (CONV) {
      tap = k0 / g.cin;
      kin = k0 - tap * g.cin;
    }
    const int shift = tap - (g.taps >> 1);
    unsigned z = 0u;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      size_t row;
      if (CONV) {
        const int s2 = a_s[p] + shift;
        const bool ok = a_ok[p] && s2 >= 0 && s2 < g.seq_len;
        if (!ok) z |= 1u << p;
        row = ok ? (size_t)a_b[p] * g.seq_len + s2 : 0;
      } else {
        row = a_ok[p] ? (size_t)a_s[p] : 0;
      }
      ra[p] = *(const x8*)(A + row * g.lda + kin + lcol);
    }
    zm = z;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      const int n = n0 + lrow + RPP * p;
      rw[p] = *(const x8*)(W + (size_t)(n < g.N ? n : g.N - 1) * g.ldw + k0 + lcol);
    }
  };
  auto store_tile = [&](const x8 (&ra)[PA], const x8 (&rw)[PW], unsigned zm, int buf) {
    T* as = As + buf * BM * BKP;
    T* ws = Ws + buf * BN * BKP;
#pragma unroll
    for (int p = 0; p < PA; ++p) *(x8*)(as + (lrow + RPP * p) * BKP + lcol) = (CONV && ((zm >> p) & 1u)) ? zero8 : ra[p];
#pragma unroll
    for (int p = 0; p < PW; ++p) *(x8*)(ws + (lrow + RPP * p) * BKP + lcol) = rw[p];
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int buf) {
    const T* as = As + buf * BM * BKP + (wm * TM + fr) * BKP + fg * 8;
    const T* ws = Ws + buf * BN * BKP + (wn * TN + fr) * BKP + fg * 8;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      x8 fa[FM], fw[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) fa[j] = *(const x8*)(as + j * 16 * BKP + ks * 32);
#pragma unroll
      for (int i = 0; i < FN; ++i) fw[i] = *(const x8*)(ws + i * 16 * BKP + ks * 32);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
    }
  };

  const int nt = kt_end - kt_begin;  // >= 1 (the launcher guarantees splitk <= k-tiles)
  const int last = kt_end - 1;
  // Every iteration issues its prefetch unconditionally (clamped to the last tile; a redundant reload of
  // the final tile is harmless) so the body stays straight-line and the waits stay counted.
  load_tile(ra0, rw0, zm0, kt_begin);
  load_tile(ra1, rw1, zm1, min(kt_begin + 1, last));
  store_tile(ra0, rw0, zm0, 0);
  __syncthreads();
  for (int i = 0; i < nt; i += 2) {
    // tile i is in LDS buffer 0, tile i+1 in flight in set 1, set 0 is free
    load_tile(ra0, rw0, zm0, min(kt_begin + i + 2, last));
    compute(0);
    store_tile(ra1, rw1, zm1, 1);
    __syncthreads();
    if (i + 1 >= nt) break;
    // tile i+1 is in LDS buffer 1, tile i+2 in flight in set 0, set 1 is free
    load_tile(ra1, rw1, zm1, min(kt_begin + i + 3, last));
    compute(1);
    store_tile(ra0, rw0, zm0, 0);
    __syncthreads();
  }

  run_epilogue<Epi, FM, FN, TM, TN>(g, acc, m0 + wm * TM, n0 + wn * TN, lane, z);
}

// ------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without passing through VGPRs or
// ds_write instructions (the register-staged kernel spends about as many LDS-issue cycles writing a stage as
// the MFMAs take to consume it).  A wave instruction fills 1 KiB = 8 rows x 128 B, lane-linear, so rows are
// unpadded; bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS chunk c of row r
// holds global 16-byte chunk c ^ ((r >> 1) & 7), and fragment reads apply the same involution.
// Conv padding / out-of-range rows cannot be zero-filled by a select any more: those lanes read a 16-byte
// zero page instead.  Two LDS stages; the stage for k-tile t+1 is in flight while tile t is multiplied.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <typename T, int BM, int BN, int NW, int ST, typename Epi, bool CONV, bool QUART = false>
__global__ __launch_bounds__(NW * 64) void gemm_glds_kernel(GemmArgs g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BK = 64;
  constexpr int WGN = NW / 2;
  constexpr int TM = BM / 2, TN = BN / WGN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;  // 1-KiB pieces (8 rows) per wave per stage
  static_assert(PA >= 1 && PW >= 1, "tile too small for this many waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As = (T*)smem_raw;             // [ST][BM][64]
  T* Ws = As + ST * BM * BK;        // [ST][BN][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x;
    const int nwg = gx * gridDim.y;
    const int id = blockIdx.x + gx * blockIdx.y;
    const int xcd = id & 7, loc = id >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = nid % gx;
    by = nid / gx;
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int z = blockIdx.z;
  const int nk_total = g.K / BK;
  const int kt_begin = (int)((long long)nk_total * z / g.splitk);
  const int kt_end = (int)((long long)nk_total * (z + 1) / g.splitk);
  const T* A = (const T*)g.A;
  const T* W = (const T*)g.W;
  const T* zero = (const T*)g_zero_page;

  // per-piece lane geometry: this lane fills LDS chunk lc of row (piece * 8 + lr) with global chunk lc ^ swz(row)
  const int lr = lane >> 3, lc = lane & 7;
  int a_row[PA], a_b[PA], a_s[PA], a_src[PA];
  bool a_ok[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = (wave + NW * p) * 8 + lr;
    a_row[p] = row;
    a_src[p] = (lc ^ ((row >> 1) & 7)) * 8;
    const int m = m0 + row;
    a_ok[p] = m < g.M;
    if (CONV) {
      a_b[p] = m / g.seq_len;
      a_s[p] = m - a_b[p] * g.seq_len;
    } else {
      a_b[p] = 0;
      a_s[p] = a_ok[p] ? m : 0;  // rows beyond M re-read row 0: their outputs are never stored
    }
  }
  const T* w_ptr[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int row = (wave + NW * p) * 8 + lr;
    const int n = n0 + row;
    if (g.w_packed) {
      // tile-packed weights [N/64][K/64][64][64]: every 64x64 k-tile of a column panel is one contiguous 8 KiB
      // block and a panel is one contiguous run, so a block streams sequential DRAM pages instead of touching
      // 64 rows that lie K*2 bytes apart (row-major streaming measured ~2 TB/s, a quarter of HBM peak).
      const int nc = n < g.n_pad ? n : g.n_pad - 1;
      w_ptr[p] = W + ((size_t)(nc >> 6) * (g.K >> 6) * 64 + (nc & 63)) * 64 + (lc ^ ((row >> 1) & 7)) * 8;
    } else {
      w_ptr[p] = W + (size_t)(n < g.N ? n : g.N - 1) * g.ldw + (lc ^ ((row >> 1) & 7)) * 8;
    }
  }
  const int w_tile_stride = g.w_packed ? 64 * 64 : BK;  // elements between consecutive k-tiles of a W row

  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
    int tap = 0, kin = k0;
    if (CONV) {
      tap = k0 / g.cin;
      kin = k0 - tap * g.cin;
    }
    const int shift = tap - (g.taps >> 1);
    T* as = As + buf * BM * BK;
    T* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const T* src;
      if (CONV) {
        const int s2 = a_s[p] + shift;
        const bool ok = a_ok[p] && s2 >= 0 && s2 < g.seq_len;
        src = ok ? A + ((size_t)a_b[p] * g.seq_len + s2) * g.lda + kin + a_src[p] : zero;
      } else {
        src = A + (size_t)a_s[p] * g.lda + kin + a_src[p];
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(as + (wave + NW * p) * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_ptr[p] + (size_t)kt * w_tile_stride), (lds_void_t*)(ws + (wave + NW * p) * 8 * BK), 16, 0, 0);
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int buf) {
    const T* as = As + buf * BM * BK;
    const T* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      x8 fa[FM], fw[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int r = wm * TM + j * 16 + fr;
        fa[j] = *(const x8*)(as + r * BK + (((ks * 4 + fg) ^ ((r >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int r = wn * TN + i * 16 + fr;
        fw[i] = *(const x8*)(ws + r * BK + (((ks * 4 + fg) ^ ((r >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
    }
  };

  if constexpr (ST == 2) {
    issue(kt_begin, 0);
    __syncthreads();  // (drains the LDS-DMA: hipcc emits vmcnt(0) before the barrier while a global_load_lds is pending)
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (kt + 1 < kt_end) issue(kt + 1, cur ^ 1);
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // ST-stage ring, ST-1 tiles in flight.  One raw barrier per k-step; the wait is a COUNTED vmcnt so the
    // newer stages stay in flight across the barrier (a __syncthreads() here would drain them: vmcnt(0)).
    // Every iteration issues exactly G loads (tile index clamped; a redundant reload targets the ring slot
    // that was consumed last iteration and is never read again), which keeps the count uniform in the tail.
    constexpr int G = PA + PW;
    const int nt = kt_end - kt_begin;
    if constexpr (!QUART) {
      const int last = kt_end - 1;
#pragma unroll
      for (int s = 0; s < ST - 1; ++s) issue(min(kt_begin + s, last), s);
      int slot = 0;
      for (int kt = kt_begin; kt < kt_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
        __builtin_amdgcn_s_barrier();
        int nslot = slot + ST - 1;
        if (nslot >= ST) nslot -= ST;
        issue(min(kt + ST - 1, last), nslot);
        compute(slot);
        slot = slot + 1 == ST ? 0 : slot + 1;
      }
    } else {
      // Decode shapes: a CU streams from HBM at only ~24 GB/s, so the four row-tile blocks that share a W panel
      // must not all pull it in the same order.  The k-range is cut into four quarters; block bx starts at quarter
      // bx & 3 and wraps, so at any moment the siblings fetch DIFFERENT quarters from HBM and find the others in L2.
      // Each quarter is summed into its own accumulator in natural k order and the four are combined in a fixed
      // order, so the result is bit-identical for every rotation (and for every batch size / sharding).
      const bool quart = (nt & 3) == 0;
      const int qlen = quart ? nt >> 2 : nt;
      const int rot = quart ? (bx & 3) : 0;
      auto tile_of = [&](int i) {  // i-th tile in this block's visiting order (clamped in the tail)
        const int ii = i < nt ? i : nt - 1;
        int t = ii + rot * qlen;
        if (t >= nt) t -= nt;
        return kt_begin + t;
      };
      f32x4 accq[4][FN][FM];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) accq[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < ST - 1; ++s) issue(tile_of(s), s);
      int slot = 0, in_q = 0, seg = 0;
      for (int i = 0; i < nt; ++i) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
        __builtin_amdgcn_s_barrier();
        int nslot = slot + ST - 1;
        if (nslot >= ST) nslot -= ST;
        issue(tile_of(i + ST - 1), nslot);
        compute(slot);
        slot = slot + 1 == ST ? 0 : slot + 1;
        if (++in_q == qlen) {  // quarter finished: bank it (block-uniform branch, 4 times per kernel)
          in_q = 0;
          const int q = quart ? ((seg + rot) & 3) : 0;
          ++seg;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            if (qq == q) {
#pragma unroll
              for (int a = 0; a < FN; ++a)
#pragma unroll
                for (int b = 0; b < FM; ++b) {
                  accq[qq][a][b] += acc[a][b];
                  acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = ((accq[0][a][b] + accq[1][a][b]) + accq[2][a][b]) + accq[3][a][b];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  run_epilogue<Epi, FM, FN, TM, TN>(g, acc, m0 + wm * TM, n0 + wn * TN, lane, z);
}

template <int BM, int BN, int ST>
constexpr int smem_bytes_glds() {
  return ST * (BM + BN) * 64 * 2;
}

template <typename T, int BM, int BN, int NW, int ST, typename Epi, bool QUART = false>
static int launch_glds(const GemmArgs& a, hipStream_t stream, int prof_tile) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.splitk);
  constexpr int smem = smem_bytes_glds<BM, BN, ST>();
  const double out_bytes = (double)a.M * a.N * ((a.out_f32 || Epi::kId != 0 ? 4.0 : 0.0) * (Epi::kId == 0 ? 1.0 : 0.0) + (a.out_t || Epi::kId != 0 ? 2.0 : 0.0));
  ProfScope ps(prof_tile * 3 + Epi::kId, stream, 2.0 * a.M * a.N * a.K,
               ((double)a.N * a.K + (double)a.M * a.cin) * 2.0 + out_bytes * (a.splitk > 1 ? a.splitk : 1) + (a.res ? 4.0 * a.M * a.N : 0.0));
  if constexpr (Epi::kId == 0) {
    if (a.taps > 1) gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, QUART><<<grid, dim3(NW * 64), smem, stream>>>(a);
    else gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, QUART><<<grid, dim3(NW * 64), smem, stream>>>(a);
  } else {
    gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, QUART><<<grid, dim3(NW * 64), smem, stream>>>(a);
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int BM, int BN, int BK>
constexpr int smem_bytes() {
  return 2 * (BM + BN) * (BK + 8) * 2;
}

// tile configurations: id -> (BM, BN, BK)
//   0: 64x64x64   1: 128x64x64   2: 128x128x64   3: 64x64x256 (decode: few fat k-stages)
//   4: 128x128x64 with 8 waves   5: 128x64x64 with 8 waves   (two waves per SIMD overlap LDS and MFMA phases)
//   6: 128x128 / 7: 128x64 (8 waves), 8: 64x64 (4 waves): direct-to-LDS (global_load_lds) staging
template <typename T, int BM, int BN, int BK, int NW, typename Epi>
static int launch_one(const GemmArgs& a, hipStream_t stream, int tile_id) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.splitk);
  constexpr int smem = smem_bytes<BM, BN, BK>();
  // algorithmic work of this launch: 2*M*N*K flops; operands read once + result written once
  const double out_bytes = (double)a.M * a.N * ((a.out_f32 || Epi::kId != 0 ? 4.0 : 0.0) * (Epi::kId == 0 ? 1.0 : 0.0) + (a.out_t || Epi::kId != 0 ? 2.0 : 0.0));
  const int prof_tile = tile_id == 3 ? 0 : (tile_id == 4 ? 2 : (tile_id == 5 ? 1 : tile_id));
  ProfScope ps(prof_tile * 3 + Epi::kId, stream, 2.0 * a.M * a.N * a.K,
               ((double)a.N * a.K + (double)a.M * a.cin) * 2.0 + out_bytes * (a.splitk > 1 ? a.splitk : 1) + (a.res ? 4.0 * a.M * a.N : 0.0));
  if constexpr (Epi::kId == 0) {
    if (a.taps > 1) gemm_kernel<T, BM, BN, BK, NW, Epi, true><<<grid, dim3(NW * 64), smem, stream>>>(a);
    else gemm_kernel<T, BM, BN, BK, NW, Epi, false><<<grid, dim3(NW * 64), smem, stream>>>(a);
  } else {
    gemm_kernel<T, BM, BN, BK, NW, Epi, false><<<grid, dim3(NW * 64), smem, stream>>>(a);
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

static int forced_tile() {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("TT_GEMM_TILE");  // experiments only (scripts/kbench.py)
    v = e ? atoi(e) : -1;
  }
  return v;
}

static int pick_tile(const GemmArgs& a) {
  int tile = forced_tile();
  if (tile < 0) {
    // Direct-to-LDS kernels by default (measured on MI355X, scripts/kbench.py):
    //   >= 256 tiles of 128x128 : 128x128, 8 waves, 2 stages (highest flop per L2 byte; 2 blocks per CU)
    //   fewer, M > 1024         : 128x64, 8 waves, 4-stage ring (more blocks, 3 tiles in flight)
    //   decode / M <= 1024      : 64x64, 4 waves, 4-stage ring (weights stream from HBM: depth hides latency)
    const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.splitk;
    // A GEMM that emits GroupNorm statistics keeps the 128x64 tile for every M > 256: the statistics are grouped per
    // wave tile, so the same tile at one and at two batch rows keeps the denoiser's conditioned row bit-identical
    // whether it is evaluated alone (split tail) or batched with the conditioning-free row.
    if (a.M > 256 && b128 >= 256) tile = 6;
    else if (a.M > 1024 || (a.M > 256 && a.gn_part != nullptr)) tile = 10;
    else tile = 11;  // also one denoiser row (M = S <= 1024, the split diffusion tail): 2x the workgroups of 128x64
  }
  if (tile == 3 && (a.cin % 256 != 0 || (a.K / 256) < a.splitk)) tile = 0;
  if (a.w_packed && tile < 6) tile = a.M > 256 ? 10 : 11;  // only the direct-to-LDS kernels read tile-packed weights
  return tile;
}

// rows per statistics tile (= the wave tile height TM of the kernel that pick_tile selects)
static int tile_stat_rows(int tile) {
  switch (tile) {
    case 0: case 3: case 8: case 11: case 14: return 32;
    default: return 64;
  }
}

template <typename T, typename Epi>
static int launch_tiles(const GemmArgs& a, hipStream_t stream) {
  const int tile = pick_tile(a);
  switch (tile) {
    case 14: return launch_glds<T, 64, 64, 4, 4, Epi, true>(a, stream, 0);
    case 11: return launch_glds<T, 64, 64, 4, 4, Epi>(a, stream, 0);
    case 10: return launch_glds<T, 128, 64, 8, 4, Epi>(a, stream, 1);
    case 9: return launch_glds<T, 128, 128, 8, 3, Epi>(a, stream, 2);
    case 8: return launch_glds<T, 64, 64, 4, 2, Epi>(a, stream, 0);
    case 7: return launch_glds<T, 128, 64, 8, 2, Epi>(a, stream, 1);
    case 6: return launch_glds<T, 128, 128, 8, 2, Epi>(a, stream, 2);
    case 5: return launch_one<T, 128, 64, 64, 8, Epi>(a, stream, 5);
    case 4: return launch_one<T, 128, 128, 64, 8, Epi>(a, stream, 4);
    case 3: return launch_one<T, 64, 64, 256, 4, Epi>(a, stream, 3);
    case 2: return launch_one<T, 128, 128, 64, 4, Epi>(a, stream, 2);
    case 1: return launch_one<T, 128, 64, 64, 4, Epi>(a, stream, 1);
    default: return launch_one<T, 64, 64, 64, 4, Epi>(a, stream, 0);
  }
}

template <typename T>
static int launch_epi(int epi, const GemmArgs& a, hipStream_t stream) {
  switch (epi) {
    case EPI_STD: return launch_tiles<T, EpiStd<T>>(a, stream);
    case EPI_QKV_HEADS: return launch_tiles<T, EpiQkvHeads<T>>(a, stream);
    case EPI_QKV_DECODE: return launch_tiles<T, EpiQkvDecode<T>>(a, stream);
  }
  set_error("gemm: unknown epilogue %d", epi);
  return -1;
}

static void normalise(GemmArgs& a) {
  if (a.taps < 1) a.taps = 1;
  if (a.splitk < 1) a.splitk = 1;
  a.cin = a.K / a.taps;
  static int xm = -2;
  if (xm == -2) {
    const char* e = getenv("TT_GEMM_XCD");
    xm = e ? atoi(e) : -1;
  }
  a.xcd_mode = xm >= 0 ? xm : 1;
}

int gemm_stat_rows(const GemmArgs& a0) {
  GemmArgs a = a0;
  normalise(a);
  return tile_stat_rows(pick_tile(a));
}

int gemm_launch(int dtype, int epi, const GemmArgs& a0, hipStream_t stream) {
  GemmArgs a = a0;
  normalise(a);
  TT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  TT_REQUIRE(a.K % 64 == 0 && a.cin % 64 == 0, "gemm: K=%d (taps=%d) must be a multiple of 64 per tap", a.K, a.taps);
  TT_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda=%d / ldw=%d must be multiples of 8 elements", a.lda, a.ldw);
  TT_REQUIRE(a.splitk == 1 || (epi == EPI_STD && a.out_f32 != nullptr), "gemm: split-K needs EPI_STD with an f32 slab output");
  TT_REQUIRE(a.splitk <= a.K / 64, "gemm: splitk=%d exceeds the %d k-tiles", a.splitk, a.K / 64);
  if (a.w_packed) {
    TT_REQUIRE(a.taps == 1, "gemm: tile-packed weights are not supported for conv taps");
    a.n_pad = (a.N + 63) / 64 * 64;
  }
  if (a.gn_part) {
    TT_REQUIRE(epi == EPI_STD && a.splitk == 1 && a.out_f32 && a.gn_seq > 0 && a.N % 16 == 0, "gemm: GroupNorm statistics need the standard epilogue, an f32 output, no split-K and N %% 16 == 0");
    a.gn_ncol16 = a.N / 16;
  }
  if (epi != EPI_STD) {
    TT_REQUIRE(a.taps == 1, "gemm: conv taps are only supported with the standard epilogue");
    TT_REQUIRE(a.dmodel % 64 == 0 && a.N == 3 * a.dmodel && a.heads * 64 == a.dmodel, "gemm: qkv epilogue needs N == 3*dmodel, head_dim 64");
  }
  if (a.taps > 1 || epi == EPI_QKV_HEADS) TT_REQUIRE(a.seq_len > 0 && a.M % a.seq_len == 0, "gemm: M=%d is not a whole number of sequences of %d", a.M, a.seq_len);
  if (dtype == DT_BF16) return launch_epi<bf16>(epi, a, stream);
  if (dtype == DT_F16) return launch_epi<f16>(epi, a, stream);
  set_error("gemm: unknown dtype %d", dtype);
  return -1;
}

template <typename T, int BM, int BN, int BK, int NW, typename Epi>
static int set_attr_one() {
  const void* fn = (const void*)gemm_kernel<T, BM, BN, BK, NW, Epi, false>;
  TT_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<BM, BN, BK>()));
  if constexpr (Epi::kId == 0) {
    const void* fc = (const void*)gemm_kernel<T, BM, BN, BK, NW, Epi, true>;
    TT_CHECK_HIP(hipFuncSetAttribute(fc, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<BM, BN, BK>()));
  }
  return 0;
}
template <typename T, int BM, int BN, int NW, int ST, typename Epi, bool QUART = false>
static int set_attr_glds() {
  const void* fn = (const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, QUART>;
  TT_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes_glds<BM, BN, ST>()));
  if constexpr (Epi::kId == 0) {
    const void* fc = (const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, QUART>;
    TT_CHECK_HIP(hipFuncSetAttribute(fc, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes_glds<BM, BN, ST>()));
  }
  return 0;
}
template <typename T, typename Epi>
static int set_attr() {
  TT_TRY((set_attr_glds<T, 128, 128, 8, 2, Epi>()));
  TT_TRY((set_attr_glds<T, 128, 64, 8, 2, Epi>()));
  TT_TRY((set_attr_glds<T, 64, 64, 4, 2, Epi>()));
  TT_TRY((set_attr_glds<T, 128, 128, 8, 3, Epi>()));
  TT_TRY((set_attr_glds<T, 128, 64, 8, 4, Epi>()));
  TT_TRY((set_attr_glds<T, 64, 64, 4, 4, Epi>()));
  TT_TRY((set_attr_glds<T, 64, 64, 4, 4, Epi, true>()));
  TT_TRY((set_attr_one<T, 128, 128, 64, 4, Epi>()));
  TT_TRY((set_attr_one<T, 128, 64, 64, 4, Epi>()));
  TT_TRY((set_attr_one<T, 64, 64, 64, 4, Epi>()));
  TT_TRY((set_attr_one<T, 64, 64, 256, 4, Epi>()));
  TT_TRY((set_attr_one<T, 128, 128, 64, 8, Epi>()));
  TT_TRY((set_attr_one<T, 128, 64, 64, 8, Epi>()));
  return 0;
}

int gemm_init() {
  TT_TRY((set_attr<bf16, EpiStd<bf16>>()));
  TT_TRY((set_attr<bf16, EpiQkvHeads<bf16>>()));
  TT_TRY((set_attr<bf16, EpiQkvDecode<bf16>>()));
  TT_TRY((set_attr<f16, EpiStd<f16>>()));
  TT_TRY((set_attr<f16, EpiQkvHeads<f16>>()));
  TT_TRY((set_attr<f16, EpiQkvDecode<f16>>()));
  return 0;
}

}  // namespace tt
