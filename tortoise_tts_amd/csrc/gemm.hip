// MFMA GEMM for gfx950 (see gemm.h).  One kernel family, gemm_glds_kernel: NW waves in a 2 x NW/2 grid, each wave owns a
// (BM/2) x (BN / (NW/2)) sub-tile built from v_mfma_f32_16x16x32 fragments.  The MFMA is issued "swapped" (W fragment as
// the A operand, activation fragment as B) so that a lane ends up holding four consecutive output columns n..n+3 of one
// row m: epilogue stores are 16 B (f32) / 8 B (bf16).  Tiles move global -> LDS directly (global_load_lds_dwordx4, no
// register stage, no ds_write), XOR-swizzled on the source side; conv taps shift the source rows per k-tile and read a
// zero page for the sequence-edge padding.  The tile is chosen from the problem shape only (pick_tile): there are no
// run-time overrides in the product library; experiments live in csrc/kbench/ (built with `build.py --kbench`).
#include "gemm.h"

#ifndef TT_EPI_FETCH
#define TT_EPI_FETCH 1  // 0: before the ring fill, 1: right after the ring fill (default), 2: after the k-loop (A/B builds only)
#endif

namespace tt {

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
  // AL (compile time): N % 4 == 0 and every operand / output row is 16-byte aligned, so every access is a whole quad
  template <bool AL>
  __device__ __forceinline__ void store(const GemmArgs& g, int m, int n, const f32x4& v, int nvalid, int z) const {
    if (g.splitk > 1) {
      float* o = g.out_f32 + (size_t)z * g.M * g.ldo32 + (size_t)m * g.ldo32 + n;
      if (AL || (nvalid == 4 && (g.ldo32 & 3) == 0)) {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
      return;
    }
    if (g.out_f32) {
      float* o = g.out_f32 + (size_t)m * g.ldo32 + n;
      if (AL || (nvalid == 4 && (g.ldo32 & 3) == 0)) {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
    }
    if (g.out_t) {
      T* o = (T*)g.out_t + (size_t)m * g.ldot + n;
      if (AL || (nvalid == 4 && (g.ldot & 3) == 0)) {
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
  template <bool AL>
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
  template <bool AL>
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

// Epilogue.  With g.gn_part set (EPI_STD, f32 output feeding a GroupNorm32) every
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

// Bias and residual operands of a wave tile, requested as whole quads BEFORE the k-loop (epi_fetch) so that their memory
// round trip (the bias vector is HBM-cold every step) overlaps the whole loop instead of sitting between the last MFMA and
// the first store (measured: 1.2 - 1.7 us per launch at the decode and denoiser shapes).  One round trip per strip, never
// one per element (a per-element `if (i < nvalid) v += bias[n + i]` compiles to load / s_waitcnt vmcnt(0) / branch
// chains).  Out-of-range rows / columns are clamped, never stored.
template <int FM, int FN>
struct EpiOperands {
  float4 bv[FN], rv[FN][FM];
};

template <typename Epi, int FM, int FN, bool AL>
__device__ __forceinline__ void epi_fetch(const GemmArgs& g, EpiOperands<FM, FN>& o, int m0w, int n0w, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
  const bool use_bias = g.bias != nullptr && !(Epi::kId == 0 && g.splitk > 1);
  const bool use_res = Epi::kId == 0 && g.res != nullptr && g.splitk == 1;
  const bool quads = AL;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0w + i * 16 + fg * 4;
    const int nvalid = g.N - n >= 4 ? 4 : g.N - n;
    o.bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < FM; ++j) o.rv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quads) {
      const int nc = max(min(n, g.N - 4), 0);
      if (use_bias) o.bv[i] = *(const float4*)(g.bias + nc);
      if (use_res) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int mc = min(m0w + j * 16 + fr, g.M - 1);
          o.rv[i][j] = *(const float4*)(g.res + (size_t)mc * g.ldres + nc);
        }
      }
    } else {
      if (use_bias) o.bv[i] = load_upto4(g.bias + n, nvalid);
      if (use_res) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int m = m0w + j * 16 + fr;
          if (m < g.M) o.rv[i][j] = load_upto4(g.res + (size_t)m * g.ldres + n, nvalid);
        }
      }
    }
  }
}

template <typename Epi, int FM, int FN, int TM, int TN, bool AL>
__device__ __forceinline__ void run_epilogue(const GemmArgs& g, f32x4 (&acc)[FN][FM], const EpiOperands<FM, FN>& o, int m0w, int n0w, int lane, int z) {
  const int fr = lane & 15, fg = lane >> 4;
  Epi epi;
  const bool stats = Epi::kId == 0 && g.gn_part != nullptr && g.splitk == 1;
  const int rt = m0w / TM;                              // row-tile index (m0w is a multiple of TM)
  const int b_first = stats ? m0w / g.gn_seq : 0;
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
      epi.apply(g, acc[i][j], o.bv[i], o.rv[i][j]);
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
      if (m < g.M && n < g.N) epi.template store<AL>(g, m, n, acc[i][j], nvalid, z);
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

// ------------------------------------------------------------------------------------------------------
// Direct-to-LDS staging (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without passing through VGPRs or
// ds_write instructions.  A wave instruction fills 1 KiB = 8 rows x 128 B, lane-linear, so rows are
// unpadded; bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS chunk c of row r
// holds global 16-byte chunk c ^ ((r >> 1) & 7), and fragment reads apply the same involution.
// Conv padding / out-of-range rows cannot be zero-filled by a select any more: those lanes read a 16-byte
// zero page instead.  Two LDS stages; the stage for k-tile t+1 is in flight while tile t is multiplied.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <typename T, int BM, int BN, int NW, int ST, typename Epi, bool CONV, bool AL>
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
  // XCD-aware tile order.  Hardware deals workgroup i to XCD i % 8, each with a private 4 MiB L2, and everything that is
  // not in the LOCAL L2 arrives over the fabric at HBM-like bandwidth (~6.5 TB/s for the whole chip, Infinity-Cache hits
  // included: scripts/kbench.py bw).  So the tile grid is cut into `xcd_rows` row bands and every XCD owns a contiguous
  // run of (band, column, row-in-band)-ordered tiles, i.e. a rectangle of about (gx / xcd_rows) x (8 gy / ... ) tiles:
  // it pulls A / xcd_rows + W * xcd_rows / 8 over the fabric instead of all of A (xcd_rows = 1, the decode shapes where
  // A is tiny) or all of W (xcd_rows = 8).  gemm_launch picks xcd_rows to minimise that sum.
  int bx, by;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nwg = gx * gy;
    const int id = blockIdx.x + gx * blockIdx.y;
    const int xcd = id & 7, loc = id >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int hb = g.xcd_band;                 // row tiles per band (the last band may be shorter)
    const int band = nid / (hb * gy);
    const int rem = nid - band * hb * gy;
    const int h = min(hb, gx - band * hb);
    by = rem / h;
    bx = band * hb + rem - by * h;
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int z = blockIdx.z;
  // split-K slab z covers k-tiles [kt_begin, kt_end): nk_total / splitk each, the first nk_total % splitk slabs one more.
  // (host-computed quotient / remainder: a 64-bit division here costs ~1 us of scalar prologue per launch)
  const int kt_begin = z * g.sk_quot + min(z, g.sk_rem);
  const int kt_end = kt_begin + g.sk_quot + (z < g.sk_rem ? 1 : 0);
  const T* A = (const T*)g.A;
  const T* W = (const T*)g.W;
  const T* zero = (const T*)g_zero_page;
  const T* A2 = nullptr;
  if (!CONV && g.A2) A2 = (const T*)g.A2 + (g.a2_slot ? (size_t)(*g.a2_slot) * g.a2_slot_stride : 0);

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
        const bool second = A2 != nullptr && kin >= g.k_split;  // block-uniform: k-tiles never straddle k_split (multiple of 64)
        src = second ? A2 + (size_t)a_s[p] * g.lda2 + (kin - g.k_split) + a_src[p] : A + (size_t)a_s[p] * g.lda + kin + a_src[p];
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
  EpiOperands<FM, FN> eo;
#if TT_EPI_FETCH == 0
  epi_fetch<Epi, FM, FN, AL>(g, eo, m0 + wm * TM, n0 + wn * TN, lane);  // requested now, consumed after the k-loop
#endif

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
#if TT_EPI_FETCH == 1
    // two-stage variant: every barrier drains the queue anyway, so the epilogue operands go out with the first tile
    // (inside the loop, even on the last iteration only, the request de-pipelines the loop: CLVP 0.033 -> 0.037 s)
    epi_fetch<Epi, FM, FN, AL>(g, eo, m0 + wm * TM, n0 + wn * TN, lane);
#endif
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
    const int last = kt_end - 1;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) issue(min(kt_begin + s, last), s);
#if TT_EPI_FETCH == 1
    // Bias / residual operands are requested AFTER the ring fill: memory operations retire in order, so a residual quad
    // requested first would have to land before the first k-step may start; here it only has to land before stage ST - 1
    // is consumed (the counted waits below over-wait by these few loads during the first two k-steps, nothing more).
    epi_fetch<Epi, FM, FN, AL>(g, eo, m0 + wm * TM, n0 + wn * TN, lane);
#endif
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#if TT_EPI_FETCH == 2
  epi_fetch<Epi, FM, FN, AL>(g, eo, m0 + wm * TM, n0 + wn * TN, lane);
#endif

  run_epilogue<Epi, FM, FN, TM, TN, AL>(g, acc, eo, m0 + wm * TM, n0 + wn * TN, lane, z);
}

template <int BM, int BN, int ST>
constexpr int smem_bytes_glds() {
  return ST * (BM + BN) * 64 * 2;
}

template <typename T, int BM, int BN, int NW, int ST, typename Epi>
static int launch_glds(const GemmArgs& a, hipStream_t stream, int prof_id) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.splitk);
  constexpr int smem = smem_bytes_glds<BM, BN, ST>();
  // algorithmic work of this launch: 2*M*N*K flops; operands read once + result written once
  const double out_bytes = (double)a.M * a.N * ((a.out_f32 || Epi::kId != 0 ? 4.0 : 0.0) * (Epi::kId == 0 ? 1.0 : 0.0) + (a.out_t || Epi::kId != 0 ? 2.0 : 0.0));
  ProfScope ps(prof_id, stream, 2.0 * a.M * a.N * a.K,
               ((double)a.N * a.K + (double)a.M * a.cin) * 2.0 + out_bytes * (a.splitk > 1 ? a.splitk : 1) + (a.res ? 4.0 * a.M * a.N : 0.0));
  // aligned fast path: whole-quad operand fetches and stores with no per-element fallback code in the kernel
  const bool al = (a.N & 3) == 0 && a.N >= 4 && (!a.bias || ((size_t)a.bias & 15) == 0) &&
                  (!a.res || (((size_t)a.res & 15) == 0 && (a.ldres & 3) == 0)) &&
                  (!a.out_f32 || (((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0)) && (!a.out_t || (((size_t)a.out_t & 7) == 0 && (a.ldot & 3) == 0));
  if constexpr (Epi::kId == 0) {
    if (a.taps > 1) {
      if (al) gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, true><<<grid, dim3(NW * 64), smem, stream>>>(a);
      else gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, false><<<grid, dim3(NW * 64), smem, stream>>>(a);
    } else {
      if (al) gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, true><<<grid, dim3(NW * 64), smem, stream>>>(a);
      else gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, false><<<grid, dim3(NW * 64), smem, stream>>>(a);
    }
  } else {
    gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, true><<<grid, dim3(NW * 64), smem, stream>>>(a);
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

enum Tile { TILE_64x64 = 0, TILE_128x64 = 1, TILE_128x128 = 2 };

// Tile choice from the problem shape only (measured on MI355X, scripts/kbench.py):
//   >= 256 tiles of 128x128 : 128x128, 8 waves, 2 stages (highest flop per L2 byte; 2 blocks per CU)
//   fewer, M > 1024         : 128x64, 8 waves, 4-stage ring (more blocks, 3 tiles in flight)
//   decode / M <= 1024      : 64x64, 4 waves, 4-stage ring (weights stream from HBM: depth hides latency)
// A GEMM that emits GroupNorm statistics keeps the 128x64 tile for every M > 256: the statistics are grouped per wave tile,
// so the same tile at one and at two batch rows keeps the denoiser's conditioned row bit-identical whether it is evaluated
// alone (split tail) or batched with the conditioning-free row.
static int pick_tile(const GemmArgs& a) {
  const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.splitk;
  if (a.M > 256 && b128 >= 256) return TILE_128x128;
  if (a.M > 1024 || (a.M > 256 && a.gn_part != nullptr)) return TILE_128x64;
  return TILE_64x64;  // also one denoiser row (M = S <= 1024, the split diffusion tail): 2x the workgroups of 128x64
}

// rows per statistics tile (= the wave tile height TM of the kernel that pick_tile selects)
static int tile_stat_rows(int tile) { return tile == TILE_64x64 ? 32 : 64; }

// ProfScope classes: (tile, epilogue, conv?) -> one class per kernel that actually runs
static int prof_class(int tile, int epi, bool conv) {
  if (epi == EPI_STD) return (tile == TILE_64x64 ? PROF_GEMM_64x64_STD : tile == TILE_128x64 ? PROF_GEMM_128x64_STD : PROF_GEMM_128x128_STD) + (conv ? 1 : 0);
  if (epi == EPI_QKV_HEADS) return tile == TILE_64x64 ? PROF_GEMM_64x64_QKV : tile == TILE_128x64 ? PROF_GEMM_128x64_QKV : PROF_GEMM_128x128_QKV;
  return tile == TILE_64x64 ? PROF_GEMM_64x64_QKVDEC : tile == TILE_128x64 ? PROF_GEMM_128x64_QKVDEC : PROF_GEMM_128x128_QKVDEC;
}

// Row bands for XCD ownership: minimise the bytes one XCD pulls over the fabric, A / bands + W * bands / 8.
static void pick_xcd_bands(GemmArgs& a, int bm) {
  const int gx = cdiv(a.M, bm);
  int xr = a.xcd_rows;
  if (xr != 1 && xr != 2 && xr != 4 && xr != 8) {
    const double A = (double)a.M * a.cin, W = (double)a.N * a.K;
    double best = 0;
    xr = 1;
    for (int c = 1; c <= 8; c *= 2) {
      const double cost = A / c + W * c / 8.0;
      if (c == 1 || cost < best) { best = cost; xr = c; }
    }
  }
  if (xr > gx) xr = gx > 0 ? gx : 1;
  a.xcd_rows = xr;
  a.xcd_band = cdiv(gx, xr);
  const int nk_total = a.K / 64;
  a.sk_quot = nk_total / a.splitk;
  a.sk_rem = nk_total % a.splitk;
}

template <typename T, typename Epi>
static int launch_tiles(const GemmArgs& a0, hipStream_t stream) {
  GemmArgs a = a0;
  const int tile = pick_tile(a);
  pick_xcd_bands(a, tile == TILE_64x64 ? 64 : 128);
  const int pc = prof_class(tile, Epi::kId, a.taps > 1);
  switch (tile) {
    case TILE_128x128: return launch_glds<T, 128, 128, 8, 2, Epi>(a, stream, pc);
    case TILE_128x64: return launch_glds<T, 128, 64, 8, 4, Epi>(a, stream, pc);
    default: return launch_glds<T, 64, 64, 4, 4, Epi>(a, stream, pc);
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
  if (a.A2) TT_REQUIRE(a.taps == 1 && a.k_split > 0 && a.k_split < a.K && a.k_split % 64 == 0 && a.lda2 % 8 == 0, "gemm: bad second activation source (k_split=%d lda2=%d)", a.k_split, a.lda2);
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

template <typename T, int BM, int BN, int NW, int ST, typename Epi>
static int set_attr_glds() {
  constexpr int smem = smem_bytes_glds<BM, BN, ST>();
  TT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  if constexpr (Epi::kId == 0) {
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_glds_kernel<T, BM, BN, NW, ST, Epi, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  }
  return 0;
}
template <typename T, typename Epi>
static int set_attr() {
  TT_TRY((set_attr_glds<T, 128, 128, 8, 2, Epi>()));
  TT_TRY((set_attr_glds<T, 128, 64, 8, 4, Epi>()));
  TT_TRY((set_attr_glds<T, 64, 64, 4, 4, Epi>()));
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
