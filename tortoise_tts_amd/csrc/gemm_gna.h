// GroupNorm-apply on the A path of a GEMM ("fused conv1d + GroupNorm + SiLU" of the denoiser: reference
// tortoise/models/diffusion_decoder.py:60-120 ResBlock in_layers / out_layers, arch_util.py:21-41 GroupNorm32).
//
// gemm_glds_kernel reads its activation operand in the operand type, written by a stand-alone gn_apply launch (f32 rows in, normalised +
// scale-shifted + SiLU'd 16-bit rows out).  Here the GEMM reads the f32 rows itself: every workgroup finalises the (sample, group)
// statistics from the producer's epilogue partials - the same fp64 fixed-order sum gn_apply makes -, folds mean / rstd / gamma / beta /
// (1 + scale) / shift into one (multiplier, offset) pair per (sample, channel) in LDS (16 KB), and on the way from the f32 rows to the LDS
// tile applies y = x * mul + off, SiLU, and the cast.  BOTH operands pass through registers here, prefetched PF k-tiles ahead (the
// register file is the ring: 16 + 16 VGPRs per stage), and are written to double-buffered LDS tiles with the swizzle of the DMA path, so
// the MFMA loop is the one of gemm_glds_kernel.  W cannot stay on the LDS-DMA: global_load_lds is a FLAT-encoded instruction that
// touches both memory and LDS, and while one is pending hipcc turns every wait it inserts for a REGISTER load into vmcnt(0) ("flat may
// return out of order") - the A rows' waits drained the whole queue every k-step (measured: +7.7 us per launch).  With plain loads only,
// the compiler's waits are counted and no hand-written vmcnt is needed.
#pragma once

namespace tt {

struct GnaArgs {
  const float* gamma;
  const float* beta;
  const float* ss;          // scale / shift rows [2C]: batch row b reads ss + (b / ss_div) * ss_stride (nullptr: none)
  size_t ss_stride;
  int ss_div;
  const float* gemm_part;   // the producing GEMM's statistics partials [row_tile][slot][C / 16][2]
  int part_shift;           // log2(rows of a statistics tile)
  int S;                    // rows per sample
  float eps;
  int act;                  // ACT_NONE / ACT_SILU
  double inv_count;         // 1 / (S * C / 32)
  int* guard;
};
template <typename EA>
struct GemmGnaDev {
  GemmCore c;
  EA e;
  GnaArgs n;
};

constexpr int kGnaC = 1024;  // channels (= K): 32 groups of 32 channels = 2 statistics strips of 16 per group


template <typename T, int BM, int BN, int NW, int WM, int PF, typename Epi, bool SS, bool SILU>
__global__ __launch_bounds__(NW * 64) void gemm_gna_kernel(const GemmGnaDev<typename Epi::Args> g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BK = 64;
  constexpr int WGN = NW / WM;
  constexpr int TM = BM / WM, TN = BN / WGN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int RP = NW * 8;        // rows of 64 elements that the workgroup's threads cover with one 8-element chunk each
  constexpr int PW = BN / RP;       // 16-byte chunks of W per thread per k-tile
  constexpr bool QUAD = BM * 8 < NW * 64;  // fewer 8-channel chunks than threads: every thread takes ONE float4 (4 channels) of the A tile instead
  constexpr int PA = QUAD ? 1 : BM / RP;   // 8-channel chunks (QUAD: 4-channel quads) of A per thread per k-tile
  static_assert((NW == 4 || NW == 8) && PW >= 1 && (QUAD ? BM * 16 == NW * 64 : BM % RP == 0), "gemm_gna: 256 or 512 threads covering the A tile exactly");
  __shared__ __attribute__((aligned(16))) T Ws[2 * BN * BK];       // [2][BN][64]
  __shared__ __attribute__((aligned(16))) T As[2 * BM * BK];       // [2][BM][64]
  __shared__ __attribute__((aligned(16))) float2 tab[2 * kGnaC];   // [2][1024] (multiplier, offset)
  __shared__ double part_s[8][32], part_q[8][32];
  __shared__ float mean_s[2][32], rstd_s[2][32];
  const GemmCore& c = g.c;
  const GnaArgs& n = g.n;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  unsigned bx, by;
  {  // XCD-aware tile order (see gemm_glds_kernel)
    const unsigned id = blockIdx.x;
    const unsigned xcd = id & 7, loc = id >> 3;
    const unsigned nid = xcd * c.xq + min(xcd, c.xr) + loc;
    unsigned rem, rr;
    const unsigned band = fdiv(nid, c.band, rem);
    const bool lastb = band == c.last_band;
    FastDiv hd;
    hd.d = lastb ? c.hlast.d : c.hfull.d;
    hd.m = lastb ? c.hlast.m : c.hfull.m;
    by = fdiv(rem, hd, rr);
    bx = band * c.hb + rr;
  }
  const int m0 = bx * BM, n0 = by * BN;
  constexpr int nk = kGnaC / 64;  // 16 k-tiles
  static_assert(nk % PF == 0, "gemm_gna: the k-loop is unrolled by the prefetch depth");
  const float* X = (const float*)c.A;
  const T* W = (const T*)c.W;
  const int S = n.S;
  unsigned r_;
  const int b0 = (int)fdiv((unsigned)m0, c.seq, r_);
  const int next_start = (b0 + 1) * S;
  const bool straddle = m0 + BM - 1 >= next_start && next_start < c.M;  // block-uniform: the tile's last rows belong to sample b0 + 1

  // this thread's A chunks: row (tid >> 3) + 32 p, channels lc * 8 .. + 7 of the k-tile
  const int lr = lane >> 3, lc = lane & 7;
  const float* a_ptr[PA];
  int a_dst[PA], a_slot[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = QUAD ? tid >> 4 : (tid >> 3) + RP * p;
    const int q = tid & 15;                // QUAD: channels q * 4 .. + 3 = half (q & 1) of 16-byte chunk q >> 1
    const int m = min(m0 + row, c.M - 1);  // rows beyond M re-read the last row: never stored, left out of the statistics
    a_ptr[p] = X + (size_t)m * c.lda + (QUAD ? q * 4 : lc * 8);
    a_dst[p] = QUAD ? row * BK + (((q >> 1) ^ ((row >> 1) & 7)) * 8) + (q & 1) * 4 : row * BK + ((lc ^ ((row >> 1) & 7)) * 8);
    a_slot[p] = (m >= next_start ? 1 : 0) * kGnaC + (QUAD ? q * 4 : lc * 8);
  }
  const T* w_ptr[PW];
  int w_dst[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int row = (tid >> 3) + RP * p;
    const int nn = n0 + row;
    w_ptr[p] = W + (size_t)(nn < c.N ? nn : c.N - 1) * c.ldw + lc * 8;
    w_dst[p] = row * BK + ((lc ^ ((row >> 1) & 7)) * 8);
  }
  float4 ra[PF][PA][2];
  x8 rw[PF][PW];
  auto issue = [&](int kt, float4 (&r)[PA][2], x8 (&w)[PW]) {  // k-tile kt -> registers (clamped: the last tiles are re-read, not used)
    const int t = min(kt, nk - 1);
#pragma unroll
    for (int p = 0; p < PW; ++p) w[p] = *(const x8*)(w_ptr[p] + t * BK);
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      r[p][0] = *(const float4*)(a_ptr[p] + t * BK);
      if constexpr (!QUAD) r[p][1] = *(const float4*)(a_ptr[p] + t * BK + 4);
    }
  };
  auto transform = [&](int kt, const float4 (&r)[PA][2], const x8 (&w)[PW]) {  // registers -> LDS tiles of k-tile kt
    const int t = min(kt, nk - 1);
    T* as = As + (kt & 1) * BM * BK;
    T* ws = Ws + (kt & 1) * BN * BK;
#pragma unroll
    for (int p = 0; p < PW; ++p) *(x8*)(ws + w_dst[p]) = w[p];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const float4* tb = (const float4*)(tab + a_slot[p] + t * BK);  // 8 (QUAD: 4) (multiplier, offset) pairs
      float xs[8] = {r[p][0].x, r[p][0].y, r[p][0].z, r[p][0].w, 0.f, 0.f, 0.f, 0.f};
      if constexpr (!QUAD) { xs[4] = r[p][1].x; xs[5] = r[p][1].y; xs[6] = r[p][1].z; xs[7] = r[p][1].w; }
      x8 o;
#pragma unroll
      for (int j = 0; j < (QUAD ? 2 : 4); ++j) {
        const float4 mo = tb[j];
        float y0 = fmaf(xs[2 * j], mo.x, mo.y), y1 = fmaf(xs[2 * j + 1], mo.z, mo.w);
#ifndef TT_GNA_NOSILU
        if (SILU) {
          y0 *= __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
          y1 *= __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
        }
#endif
        o[2 * j] = (T)y0;
        o[2 * j + 1] = (T)y1;
      }
      if constexpr (QUAD) {
        typename Vec<T>::x4 o4;
        o4[0] = o[0]; o4[1] = o[1]; o4[2] = o[2]; o4[3] = o[3];
        *(typename Vec<T>::x4*)(as + a_dst[p]) = o4;
      } else {
        *(x8*)(as + a_dst[p]) = o;
      }
    }
  };

  // ---- requests, in need order: the first PF k-tiles, epilogue operands, affine parameters, statistics
#pragma unroll
  for (int s = 0; s < PF; ++s) issue(s, ra[s], rw[s]);
  typename Epi::template Ops<FM, FN> eo;
  Epi::template fetch<FM, FN, true>(c, g.e, eo, m0 + wm * TM, n0 + wn * TN, lane);
  const int t8 = tid & 255;  // the statistics / table code is laid out for 256 threads: with 512 the upper half repeats the lower half's work (same values)
  const int ch = t8 * 4;     // this thread's four channels of the (multiplier, offset) table
  const float4 gm = *(const float4*)(n.gamma + ch);
  const float4 bt = *(const float4*)(n.beta + ch);
  float4 sc[2], sh[2];
  sc[0] = sc[1] = sh[0] = sh[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (SS) {
    const int div = n.ss_div > 0 ? n.ss_div : 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* row = n.ss + (size_t)((b0 + (straddle ? s : 0)) / div) * n.ss_stride;
      sc[s] = *(const float4*)(row + ch);
      sh[s] = *(const float4*)(row + kGnaC + ch);
    }
  }
  // statistics of sample b0 (and b0 + 1 for the one tile that straddles): gn_finalize<1>'s sum, item by item in the same order
  {
    const int gq = t8 & 31, part = t8 >> 5;
    const int r_shift = n.part_shift, nc16 = kGnaC >> 4;
    const int nsamp = straddle ? 2 : 1;
    for (int s = 0; s < nsamp; ++s) {
      const int b = b0 + s;
      const int t0 = (b * S) >> r_shift, t1 = ((b + 1) * S - 1) >> r_shift;
      const int nitems = (t1 - t0 + 1) << 1;
      double su = 0.0, qu = 0.0;
#ifdef TT_GNA_NOSTATS
      su = 1.0; qu = 2.0 / n.inv_count;
      if (false)
#endif
      for (int e0 = part; e0 < nitems; e0 += 64) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // eight independent requests, clamped
          const int ec = min(e0 + 8 * k, nitems - 1);
          const int t = t0 + (ec >> 1), strip = (gq << 1) + (ec & 1);
          const int slot = ((t << r_shift) < b * S) ? 1 : 0;
          v[k] = *(const float2*)(n.gemm_part + (((size_t)t * 2 + slot) * nc16 + strip) * 2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (e0 + 8 * k < nitems) {
            su += (double)v[k].x;
            qu += (double)v[k].y;
          }
        }
      }
      if (s) __syncthreads();  // (part_s / part_q of the first sample have been read)
      part_s[part][gq] = su;
      part_q[part][gq] = qu;
      __syncthreads();
      if (tid < 32) {
        double ss_ = 0.0, qq = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          ss_ += part_s[p][tid];
          qq += part_q[p][tid];
        }
        const double mu = ss_ * n.inv_count;
        double var = qq * n.inv_count - mu * mu;
        if (n.guard && !(var < 1.0e300)) atomicAdd(n.guard, 1);  // NaN / inf statistics: an operand overflowed upstream
        if (var < 0.0) var = 0.0;
        mean_s[s][tid] = (float)mu;
        rstd_s[s][tid] = rsqrtf((float)var + n.eps);
      }
    }
    __syncthreads();
    for (int s = 0; s < nsamp; ++s) {
      const float mu = mean_s[s][t8 >> 3], rs = rstd_s[s][t8 >> 3];
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
      const float scv[4] = {sc[s].x, sc[s].y, sc[s].z, sc[s].w}, shv[4] = {sh[s].x, sh[s].y, sh[s].z, sh[s].w};
      float4 o[2];
      float* of = (float*)o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float mul = rs * gmv[i];
        float off = btv[i] - mu * mul;
        if (SS) {
          mul *= 1.f + scv[i];
          off = off * (1.f + scv[i]) + shv[i];
        }
        of[2 * i] = mul;
        of[2 * i + 1] = off;
      }
      *(float4*)(tab + s * kGnaC + ch) = o[0];
      *(float4*)(tab + s * kGnaC + ch + 2) = o[1];
    }
    __syncthreads();
  }
  transform(0, ra[0], rw[0]);
  issue(PF, ra[0], rw[0]);

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int kt) {
    const T* as = As + (kt & 1) * BM * BK;
    const T* ws = Ws + (kt & 1) * BN * BK;
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

  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kt = kt0 + u;
      // raw barrier (a __syncthreads() would drain the prefetch queue): the tiles of k-tile kt were written during iteration kt - 1, every
      // wave has finished reading the tiles of k-tile kt - 1, which iteration kt overwrites with those of kt + 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      transform(kt + 1, ra[(u + 1) % PF], rw[(u + 1) % PF]);
      issue(kt + 1 + PF, ra[(u + 1) % PF], rw[(u + 1) % PF]);
      compute(kt);
    }
  }
  run_epilogue<Epi, FM, FN, TM, TN, true>(c, g.e, acc, eo, eo.step(), m0 + wm * TM, n0 + wn * TN, lane, 0);
}

}  // namespace tt
