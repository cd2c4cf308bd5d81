// 256 x 256 tile on 8 waves of 256 VGPRs, eight-phase schedule (round 5; included by gemm_impl.h after gemm_glds_kernel).
//
// The 16-wave 256 x 256 tile (gemm_glds_kernel<256, 256, 16, 4, 2>) has two 64 KB stages and a draining barrier per k-tile: the next
// tile's LDS-DMA has to LAND inside one k-step of MFMA work (0.86 us at the matrix peak; the loaded latency is longer) and nothing
// overlaps the fragment reads of a wave with the MFMAs of another.  This kernel follows the CDNA4 guide's 8-phase template
// (cdna_hip_programming.md, "The 256^2 8-phase template"), schedule re-derived here:
//
//   * 8 waves as 2 (wr) x 4 (wc); LDS = 2 k-tile parities x 4 half-tiles of [128 rows][64 k] (16 KB each, the swizzled 128-byte rows of
//     gemm_glds_kernel): j = 0 W rows 0-127 ("W-lo"), 1 A rows 0-127 ("A-lo"), 2 W-hi, 3 A-hi.
//   * a wave's 128 x 64 outputs are INTERLEAVED: rows wr*64 .. +63 of A-lo and of A-hi, columns wc*32 .. +31 of W-lo and of W-hi - four
//     64 x 32 quadrants, one per phase, each reading ONE A half-tile and ONE W half-tile:
//       phase 0: read W-lo (4 fragments) + A-lo (8), quadrant (A-lo, W-lo)      phase 2: read A-hi (8), quadrant (A-hi, W-hi)
//       phase 1: read W-hi (4),                      quadrant (A-lo, W-hi)      phase 3: no reads,      quadrant (A-hi, W-lo)
//     so a half-tile's last LDS read is in phase 0 / 0 / 1 / 2 (W-lo, A-lo, W-hi, A-hi) and every buffer is re-staged >= 2 phases later.
//   * every phase issues ONE half-tile (2 LDS-DMA pieces per wave): global phase q = 4 kt + p issues half-tile h = q + 6 (h = 4 kt' + j),
//     then waits vmcnt(8) - four half-tiles stay in flight, half-tiles <= q + 2 have landed - crosses the first barrier, retires its
//     fragment reads, issues its 16 MFMAs at raised priority and crosses the second barrier.  A half-tile is read one phase after the
//     wait that retired it (phase q + 1 needs exactly h <= q + 2), never in the same phase.
//   * the two wave groups wr = 0 / 1 run half a phase apart (wr = 1 takes one extra barrier before the loop, wr = 0 one after it): while
//     one group's MFMAs run, the other group - one wave of each per SIMD - reads its fragments and issues the DMA.
//   * operands arrive through buffer_load ... lds (SGPR resource + 32-bit lane offset + the k offset in an SGPR: no address VALU per
//     phase); NO register-destination global load before or inside the loop (a pending LDS-DMA turns hipcc's waits for such loads
//     into vmcnt(0)): the epilogue operands are requested after the loop, one quadrant ahead of their use.
//
// Accumulation order per output element is the one of gemm_glds_kernel (k-tiles ascending, the two 32-wide k-steps of a tile in order):
// the results are bit-identical to the 16-wave tile's.  Epilogues: run_epilogue per quadrant with 64 x 32 wave tiles.
// Restrictions (else the 16-wave tile runs): 1 x 1 (no taps), one K source, no split-K, an even number of k-tiles, 16-byte aligned rows.
#pragma once

namespace tt {

template <typename T, typename Epi>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const GemmDev<typename Epi::Args> g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HT = 128 * BK;  // elements of a half-tile
  static_assert(!Epi::kSerial, "gemm_p8: plain epilogues only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* L = (T*)smem_raw;  // [2][4][128][64]
  const GemmCore& c = g.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
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
  const int nk = c.sk_quot;  // (no split-K: all k-tiles)

  // LDS-DMA geometry: a wave fills pieces `wave` and `wave + 8` (8 rows of 128 bytes each) of every half-tile; lane -> (row lr, chunk lc),
  // LDS chunk lc of row r holds global chunk lc ^ ((r >> 1) & 7).  Rows beyond M / N re-read row 0 / N - 1 (never stored).
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)c.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)c.W, 0, 0x7fffffff, 0x00020000);
  const int lr = lane >> 3, lc = lane & 7;
  int offA[2][2], offW[2][2];  // [half][piece] byte offsets of this lane's 16-byte chunk at k = 0
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wave + 8 * p) * 8 + lr;
    const int sw = (lc ^ ((row >> 1) & 7)) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + h * 128 + row, n = n0 + h * 128 + row;
      offA[h][p] = ((m < c.M ? m : 0) * c.lda + sw) * 2;
      offW[h][p] = ((n < c.N ? n : c.N - 1) * c.ldw + sw) * 2;
    }
  }
  auto issue = [&](int d, int j, int kt) {  // half-tile j of k-tile kt -> parity d (d, j compile-time after unrolling)
    const int so = min(kt, nk - 1) * (BK * 2);  // beyond the end: a dead buffer is re-filled with the last k-tile (uniform DMA counts)
    T* dst = L + (d * 4 + j) * HT;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      lds_void_t* lp = (lds_void_t*)(dst + (wave + 8 * p) * 8 * BK);
      if (j & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lp, 16, offA[j >> 1][p], so, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lp, 16, offW[j >> 1][p], so, 0, 0);
    }
  };

  f32x4 acc[4][2][4];  // [quadrant = 2 * (A half) + (W half)][16-column strip][16-row tile]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  x8 fa[4][2];      // [row tile][k-step] of the current A half
  x8 fw[2][2][2];   // [W half][strip][k-step]
  const int fr = lane & 15, fg = lane >> 4;
  // fragment addresses: row r of a half-tile at r * 64 elements, 16-byte chunk (ks * 4 + fg) ^ ((r >> 1) & 7); r = 16 t + fr + 64 wr (A) or
  // 16 t + fr + 32 wc (W), so (r >> 1) & 7 == fr >> 1 for every tile t
  const int ch0 = ((0 * 4 + fg) ^ (fr >> 1)) * 8, ch1 = ((1 * 4 + fg) ^ (fr >> 1)) * 8;
  const int a_row = (wr * 64 + fr) * BK, w_row = (wc * 32 + fr) * BK;
  auto read_a = [&](int d, int half) {
    const T* s = L + (d * 4 + 1 + 2 * half) * HT + a_row;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[t][0] = *(const x8*)(s + t * 16 * BK + ch0);
      fa[t][1] = *(const x8*)(s + t * 16 * BK + ch1);
    }
  };
  auto read_w = [&](int d, int half) {
    const T* s = L + (d * 4 + 2 * half) * HT + w_row;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fw[half][t][0] = *(const x8*)(s + t * 16 * BK + ch0);
      fw[half][t][1] = *(const x8*)(s + t * 16 * BK + ch1);
    }
  };
  auto quadrant = [&](int q, int wh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][i][j] = mfma16(fw[wh][i][ks], fa[j][ks], acc[q][i][j]);
  };
#if defined(TT_P8_NO_LGKM0)   // A/B knob: leave the fragment-read waits to the compiler's counted lgkmcnt in front of each MFMA
#define TT_P8_LGKM0
#else
#define TT_P8_LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#define TT_P8_SYNC_MFMA(q, wh)                          \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                         \
  TT_P8_LGKM0                                           \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_setprio(1);                        \
  quadrant(q, wh);                                      \
  __builtin_amdgcn_s_setprio(0);                        \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_barrier();                         \
  __builtin_amdgcn_sched_barrier(0);
  // the four phases of k-tile kt (parity d): reads, one half-tile issue (h = q + 6), wait, barrier, MFMAs, barrier
#define TT_P8_KTILE(d, kt)                                                          \
  read_w(d, 0);                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                \
  read_a(d, 0);                                                                     \
  issue((d) ^ 1, 2, (kt) + 1);                                                      \
  TT_P8_SYNC_MFMA(0, 0)                                                             \
  read_w(d, 1);                                                                     \
  issue((d) ^ 1, 3, (kt) + 1);                                                      \
  TT_P8_SYNC_MFMA(1, 1)                                                             \
  read_a(d, 1);                                                                     \
  issue(d, 0, (kt) + 2);                                                            \
  TT_P8_SYNC_MFMA(3, 1)                                                             \
  issue(d, 1, (kt) + 2);                                                            \
  TT_P8_SYNC_MFMA(2, 0)

  // prologue: half-tiles 0 .. 5 (k-tile 0 whole, W-lo / A-lo of k-tile 1); 0 and 1 have landed after vmcnt(8)
  issue(0, 0, 0);
  issue(0, 1, 0);
  issue(0, 2, 0);
  issue(0, 3, 0);
  issue(1, 0, 1);
  issue(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs half a phase behind the first
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    TT_P8_KTILE(0, kt)
    TT_P8_KTILE(1, kt + 1)
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the surplus re-fills of the tail)
#undef TT_P8_KTILE
#undef TT_P8_SYNC_MFMA
#undef TT_P8_LGKM0

  // epilogue, quadrant by quadrant.  The fragment registers are dead: the operands (bias, skip quads) of TT_P8_EPI_DEPTH quadrants are requested
  // before the first one is worked on and each finished quadrant's slot is re-used for the quadrant DEPTH ahead.
#ifndef TT_P8_EPI_DEPTH
#define TT_P8_EPI_DEPTH 2
#endif
  constexpr int ED = TT_P8_EPI_DEPTH;
  typename Epi::template Ops<4, 2> eo[ED];
  auto qm = [&](int q) { return m0 + (q >> 1) * 128 + wr * 64; };
  auto qn = [&](int q) { return n0 + (q & 1) * 128 + wc * 32; };
#pragma unroll
  for (int q = 0; q < ED && q < 4; ++q) Epi::template fetch<4, 2, true>(c, g.e, eo[q], qm(q), qn(q), lane);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    run_epilogue<Epi, 4, 2, 64, 32, true>(c, g.e, acc[q], eo[q % ED], eo[q % ED].step(), qm(q), qn(q), lane, 0);
    if (q + ED < 4) Epi::template fetch<4, 2, true>(c, g.e, eo[q % ED], qm(q + ED), qn(q + ED), lane);
  }
}

}  // namespace tt
