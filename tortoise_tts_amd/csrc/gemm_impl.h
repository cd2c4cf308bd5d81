// MFMA GEMM kernels for gfx950 (see gemm.h; instantiated per operand type by gemm_bf16.hip / gemm_f16.hip).
//
// One kernel family, gemm_glds_kernel: NW waves in a WM x NW/WM grid, each wave owns a (BM/WM) x (BN / (NW/WM)) sub-tile built
// from v_mfma_f32_16x16x32 fragments.  The MFMA is issued "swapped" (W fragment as the A operand, activation fragment as B)
// so that a lane ends up holding four consecutive output columns n..n+3 of one row m: epilogue stores are 16 B (f32) /
// 8 B (bf16).  Tiles move global -> LDS directly (global_load_lds_dwordx4, no register stage, no ds_write), XOR-swizzled on
// the source side; conv taps shift the source rows per k-tile and read a zero page for the sequence-edge padding.
//
// What the kernel is paid for at this engine's shapes (M = 256 decode rows, M = 1740 denoiser rows) is latency, not
// bandwidth: a launch lasts 5 - 30 us and ~80 000 of them make one utterance, so everything between the first instruction
// and the first tile request is on the critical path of every launch.  Hence:
//   * the device-side argument block (GemmDev) is compact, hot fields first, so the prologue is ONE scalar-memory round
//     trip instead of a chain of dependent kernarg loads;
//   * every integer division in the prologue / k-loop / epilogue is a multiply-high by a host-computed reciprocal;
//   * the features that a launch does not use (activation switch, GroupNorm statistics, second activation source) are
//     compiled out in the "fast" instantiations (template value 0 / 1); the value -1 keeps the run-time test (generic
//     kernel: odd alignments, rare activations and output combinations).
#pragma once
#include <type_traits>
#include "gemm.h"

namespace tt {

// ---------------------------------------------------------------------------------------------- reciprocal division
struct FastDiv {
  unsigned d, m;  // m = floor(2^32 / d) (0xFFFFFFFF for d == 1): q = mulhi(n, m) is floor(n / d) or one less
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d ? d : 1u;
  f.m = f.d == 1u ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / f.d);
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f, unsigned& rem) {
  unsigned q = __umulhi(n, f.m);
  unsigned r = n - q * f.d;
  const bool up = r >= f.d;
  q += up ? 1u : 0u;
  rem = up ? r - f.d : r;
  return q;
}

// ---------------------------------------------------------------------------------------------- device argument block
struct GemmCore {
  const void* A;           // 0
  const void* W;           // 8
  int lda, ldw;            // 16
  int M, N;                // 24
  int cin_tiles;           // 32  k-tiles per conv tap (= all k-tiles for a plain GEMM)
  int taps_half;           // 36  taps / 2
  int seq_len;             // 40
  int dil;                 // 44  rows between conv taps
  FastDiv seq;             // 48  / seq_len
  unsigned xq, xr;         // 56  workgroups / 8, workgroups % 8
  unsigned gx, gy;         // 64  row tiles, column tiles
  unsigned hb, last_band;  // 72  row tiles per XCD band, index of the last band
  FastDiv band;            // 80  / (hb * gy)
  FastDiv hfull;           // 88  / hb
  FastDiv hlast;           // 96  / (row tiles of the last band)
  int sk_quot, sk_rem;     // 104 k-tiles per split-K slab (quotient, remainder)
  int pad1, pad2;          // 112
  // second activation source (HA2): k-tiles >= a2_tile read A2
  const void* A2;          // 120
  const int* a2_slot;      // 128
  size_t a2_slot_stride;   // 136
  int lda2, a2_tile;       // 144
};

struct EpiStdArgs {
  const float* bias;
  const float* res;
  float* out_f32;
  void* out_t;
  float* gn_part;
  int ldres, ldo32, ldot, act;
  float slope;
  int splitk;
  int act_t;
  float slope_t;
  int gn_ncol16;
  FastDiv gn_seq;
  int gn_vperiod;   // padded batches (cold tail of the block: only the statistics epilogue of such a launch reads it)
  int gn_vlen[32];
};
struct EpiQkvHeadsArgs {
  const float* bias;
  void* q; void* k; void* v; void* vt;
  int heads, seq_pad;
  float q_scale;
  FastDiv dmodel;
};
struct EpiQkvDecodeArgs {
  const float* bias;
  const int* step;
  void* qbuf; void* kc; void* vc;
  int heads, tmax, dmodel_i;
  float q_scale;
  FastDiv dmodel;
};
template <typename EA>
struct GemmDev {
  GemmCore c;
  EA e;
};

__device__ __forceinline__ float4 load_upto4(const float* p, int nvalid) {  // ragged / unaligned edge: element loads
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

// Every epilogue is split in three so that the kernel can request the operands early (fetch: right behind the ring fill),
// do ALL arithmetic in registers (apply) and issue ALL stores back to back (store).
//   ACT   : -1 run-time switch on e.act, otherwise the compile-time activation
//   STATS : -1 run-time test of e.gn_part, 0 never, 1 always (GroupNorm partial statistics, see run_epilogue)
//   MODE  : -1 run-time tests of bias / res / out_f32 / out_t / splitk, otherwise the compile-time set of EB_* bits.  At
//           one wave per SIMD every instruction of the epilogue is serial latency (~3 ns each): the run-time form costs
//           ~150 instructions of flag tests, exec masking and repeated 64-bit address arithmetic per wave.
enum EpiBits { EB_BIAS = 1, EB_RES = 2, EB_F32 = 4, EB_T = 8, EB_SLAB = 16 };
template <typename T, int ACT, int STATS, int MODE>
struct EpiStd {
  typedef EpiStdArgs Args;
  static constexpr int kId = 0;
  static constexpr int kStats = STATS;
  static constexpr bool kSerial = false;
  template <int FM, int FN> struct Ops { float4 bv[FN], rv[FN][FM]; __device__ __forceinline__ int step() const { return 0; } };
  static __device__ __forceinline__ bool slab(const Args& e) { return MODE < 0 ? e.splitk > 1 : (MODE & EB_SLAB) != 0; }
  static __device__ __forceinline__ bool has_bias(const Args& e) { return MODE < 0 ? (e.bias != nullptr && e.splitk <= 1) : (MODE & EB_BIAS) != 0; }
  static __device__ __forceinline__ bool has_res(const Args& e) { return MODE < 0 ? (e.res != nullptr && e.splitk <= 1) : (MODE & EB_RES) != 0; }
  static __device__ __forceinline__ bool has_f32(const Args& e) { return MODE < 0 ? e.out_f32 != nullptr : (MODE & EB_F32) != 0; }
  static __device__ __forceinline__ bool has_t(const Args& e) { return MODE < 0 ? e.out_t != nullptr : (MODE & EB_T) != 0; }

  // Wave tiles of more than 8 fragments (the 256 x 256 tile: 16 waves per CU = 128 VGPRs each) cannot hold the prefetched skip
  // quads next to the accumulators: there the skip is read in the epilogue (run_epilogue), where the fragment registers are free
  // and a k-loop of >= 12 steps has long amortised the extra round trip.
  template <int FM, int FN> static constexpr bool late_res() { return FM * FN > 8; }
  template <bool AL>
  static __device__ __forceinline__ float4 load_res(const GemmCore& c, const Args& e, int m, int n, int nvalid) {
    if (AL) return *(const float4*)(e.res + (size_t)min(m, c.M - 1) * e.ldres + max(min(n, c.N - 4), 0));
    return m < c.M ? load_upto4(e.res + (size_t)m * e.ldres + n, nvalid) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  template <int FM, int FN, bool AL>
  static __device__ __forceinline__ void fetch(const GemmCore& c, const Args& e, Ops<FM, FN>& o, int m0w, int n0w, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    const bool use_bias = has_bias(e), use_res = has_res(e) && !late_res<FM, FN>();
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0w + i * 16 + fg * 4;
      const int nvalid = c.N - n >= 4 ? 4 : c.N - n;
      o.bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!late_res<FM, FN>()) {
#pragma unroll
        for (int j = 0; j < FM; ++j) o.rv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (AL) {
        const int nc = max(min(n, c.N - 4), 0);
        if (use_bias) o.bv[i] = *(const float4*)(e.bias + nc);
        if (use_res) {
#pragma unroll
          for (int j = 0; j < FM; ++j) {
            const int mc = min(m0w + j * 16 + fr, c.M - 1);
            o.rv[i][j] = *(const float4*)(e.res + (size_t)mc * e.ldres + nc);
          }
        }
      } else {
        if (use_bias) o.bv[i] = load_upto4(e.bias + n, nvalid);
        if (use_res) {
#pragma unroll
          for (int j = 0; j < FM; ++j) {
            const int m = m0w + j * 16 + fr;
            if (m < c.M) o.rv[i][j] = load_upto4(e.res + (size_t)m * e.ldres + n, nvalid);
          }
        }
      }
    }
  }
  static __device__ __forceinline__ void apply(const Args& e, f32x4& v, const float4& bv, const float4& rv) {
    if (slab(e)) return;  // raw partial sums; bias / activation / residual belong to the slab consumer
    if (MODE < 0 || (MODE & EB_BIAS)) { v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
    if (ACT < 0) {
      if (e.act != ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], e.act, e.slope);
      }
    } else if (ACT != ACT_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], ACT, e.slope);
    }
    if (MODE < 0 || (MODE & EB_RES)) { v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
  }
  // AL (compile time): N % 4 == 0 and every operand / output row is 16-byte aligned, so every access is a whole quad
  template <bool AL>
  static __device__ __forceinline__ void store(const GemmCore& c, const Args& e, int, int m, int n, const f32x4& v, int nvalid, int z) {
    if (slab(e)) {
      float* o = e.out_f32 + (size_t)z * c.M * e.ldo32 + (size_t)m * e.ldo32 + n;
      if (AL || (nvalid == 4 && (e.ldo32 & 3) == 0)) {
        // PLAIN stores.  Round 3 wrote the slabs through (`global_store_dwordx4 ... sc1`, -0.7 % on the decode step: no dirty lines
        // for the kernel boundary to flush).  Round 4 measured that this is only safe while the engine has the chip to itself: with a
        // second queue keeping the memory system busy (another handle on another stream, or the row ranges of tt_ar_set_option) the
        // row norm of the NEXT launch intermittently summed stale slab values - 7 of 160 generations differed in 1 - 8 rows with
        // write-through stores, 0 of 80 with plain ones, same binary otherwise (profiles/r04_concurrency_bisect.txt).  The kernel
        // boundary's release covers dirty L2 lines; it evidently does not wait for write-through traffic still on its way.
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
      return;
    }
    if (has_f32(e)) {
      float* o = e.out_f32 + (size_t)m * e.ldo32 + n;
      if (AL || (nvalid == 4 && (e.ldo32 & 3) == 0)) {
#if defined(TT_WT_F32)
        // A/B knob (build.py --variant wt -DTT_WT_F32; profiles/r05_ab_writethrough_f32.txt): the f32 result written THROUGH (sc1), so that
        // the kernel boundary finds no dirty lines to flush; every wave drains its stores at the end of run_epilogue
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(v) : "memory");
#else
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
#endif
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = v[i];
      }
    }
    if (has_t(e)) {
      T* o = (T*)e.out_t + (size_t)m * e.ldot + n;
      f32x4 w = v;
      if (MODE < 0 && e.act_t == ACT_LRELU) {  // (run-time-output variants only)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = w[i] > 0.f ? w[i] : w[i] * e.slope_t;
      }
      if (AL || (nvalid == 4 && (e.ldot & 3) == 0)) {
        *(typename Vec<T>::x4*)o = pack4<T>(w[0], w[1], w[2], w[3]);
      } else {
        for (int i = 0; i < nvalid; ++i) o[i] = (T)w[i];
      }
    }
  }
};

// N == 3 * dmodel, dmodel % 64 == 0: a wave's columns [n0w, n0w + TN), TN <= 64, never straddle a part or a head, so
// (part, head) are wave-uniform per 16-column strip and only the row -> (batch, position) split is per lane.
template <typename T>
struct EpiQkvHeads {
  typedef EpiQkvHeadsArgs Args;
  static constexpr int kId = 1;
  static constexpr int kStats = 0;
  static constexpr bool kSerial = false;
  template <int FM, int FN> struct Ops { float4 bv[FN]; __device__ __forceinline__ int step() const { return 0; } };
  template <int FM, int FN, bool AL>
  static __device__ __forceinline__ void fetch(const GemmCore& c, const Args& e, Ops<FM, FN>& o, int, int n0w, int lane) {
    const int fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int nc = max(min(n0w + i * 16 + fg * 4, c.N - 4), 0);
      o.bv[i] = e.bias ? *(const float4*)(e.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  static __device__ __forceinline__ void apply(const Args&, f32x4& v, const float4& bv, const float4&) {
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  template <bool AL>
  static __device__ __forceinline__ void store(const GemmCore& c, const Args& e, int, int m, int n, const f32x4& v, int, int) {
    unsigned cc, s;
    const unsigned part = fdiv((unsigned)n, e.dmodel, cc);
    const int h = cc >> 6, d = cc & 63;
    const unsigned b = fdiv((unsigned)m, c.seq, s);
    const size_t bh = (size_t)b * e.heads + h;
    if (part == 0) {
      T* o = (T*)e.q + (bh * c.seq_len + s) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0] * e.q_scale, v[1] * e.q_scale, v[2] * e.q_scale, v[3] * e.q_scale);
    } else if (part == 1) {
      T* o = (T*)e.k + (bh * c.seq_len + s) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    } else {
      if (e.v) {
        T* o = (T*)e.v + (bh * c.seq_len + s) * 64 + d;
        *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
      }
      if (e.vt) {
        T* o = (T*)e.vt + (bh * 64 + d) * e.seq_pad + s;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[(size_t)i * e.seq_pad] = (T)v[i];
      }
    }
  }
};

template <typename T>
struct EpiQkvDecode {
  typedef EpiQkvDecodeArgs Args;
  static constexpr int kId = 2;
  static constexpr int kStats = 0;
  static constexpr bool kSerial = false;
  template <int FM, int FN> struct Ops { float4 bv[FN]; int t; __device__ __forceinline__ int step() const { return t; } };
  template <int FM, int FN, bool AL>
  static __device__ __forceinline__ void fetch(const GemmCore& c, const Args& e, Ops<FM, FN>& o, int, int n0w, int lane) {
    const int fg = lane >> 4;
    o.t = *e.step;  // KV slot of this step (device-side counter: the captured graph is step-invariant)
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int nc = max(min(n0w + i * 16 + fg * 4, c.N - 4), 0);
      o.bv[i] = e.bias ? *(const float4*)(e.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  static __device__ __forceinline__ void apply(const Args&, f32x4& v, const float4& bv, const float4&) {
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  template <bool AL>
  static __device__ __forceinline__ void store(const GemmCore&, const Args& e, int t, int m, int n, const f32x4& v, int, int) {
    unsigned cc;
    const unsigned part = fdiv((unsigned)n, e.dmodel, cc);
    const int h = cc >> 6, d = cc & 63;
    const size_t bh = (size_t)m * e.heads + h;
    if (part == 0) {
      T* o = (T*)e.qbuf + (size_t)m * e.dmodel_i + cc;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0] * e.q_scale, v[1] * e.q_scale, v[2] * e.q_scale, v[3] * e.q_scale);
    } else if (part == 1) {
      T* o = (T*)e.kc + ((bh * 8 + (d >> 3)) * e.tmax + t) * 8 + (d & 7);
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    } else {
      T* o = (T*)e.vc + (bh * e.tmax + t) * 64 + d;
      *(typename Vec<T>::x4*)o = pack4<T>(v[0], v[1], v[2], v[3]);
    }
  }
};

// GEGLU projection: value / gate strips interleaved (gemm.h EPI_GEGLU); the product value * gelu_erf(gate) is formed in registers.
struct EpiGegluArgs {
  const float* bias;
  void* out_t;
  int ldot;
};
template <typename T>
struct EpiGeglu {
  typedef EpiGegluArgs Args;
  static constexpr int kId = 3;
  static constexpr int kStats = 0;
  static constexpr bool kSerial = false;
  template <int FM, int FN> struct Ops { float4 bv[FN]; __device__ __forceinline__ int step() const { return 0; } };
  template <int FM, int FN, bool AL>
  static __device__ __forceinline__ void fetch(const GemmCore& c, const Args& e, Ops<FM, FN>& o, int, int n0w, int lane) {
    const int fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int nc = max(min(n0w + i * 16 + fg * 4, c.N - 4), 0);
      o.bv[i] = e.bias ? *(const float4*)(e.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  static __device__ __forceinline__ void apply(const Args&, f32x4& v, const float4& bv, const float4&) {
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  // value fragment v (strip 2p), gate fragment g (strip 2p + 1), both with their bias added; n = first column of the VALUE quad
  static __device__ __forceinline__ void store_pair(const Args& e, int m, int n, const f32x4& v, const f32x4& g) {
    const int col = ((n >> 5) << 4) + (n & 15);  // strip pair p = n / 32 -> output columns 16 p ..
    T* o = (T*)e.out_t + (size_t)m * e.ldot + col;
    *(typename Vec<T>::x4*)o = pack4<T>(v[0] * gelu_erf(g[0]), v[1] * gelu_erf(g[1]), v[2] * gelu_erf(g[2]), v[3] * gelu_erf(g[3]));
  }
  template <bool AL>
  static __device__ __forceinline__ void store(const GemmCore&, const Args&, int, int, int, const f32x4&, int, int) {}
};

// "Serial split-K": out = (((res + bias) + P0) + P1) + ... with P_q the partial product over the q-th of e.splitk equal K ranges,
// each accumulated from zero in k order and folded into the running value in ONE launch.  That is bit for bit what the split-K
// slab path computes in two kernels (slab z = P_z, then the LayerNorm kernel's x + bias + slab0 + slab1 + ...), without writing
// and re-reading splitk f32 slabs: the decode projections of a batch of >= 1024 sequences (several utterances per decode batch),
// where one block per output tile already fills the chip.  The candidates' codes stay independent of the batch size.
template <typename T>
struct EpiSerial : EpiStd<T, ACT_NONE, 0, EB_BIAS | EB_RES | EB_F32> {
  static constexpr bool kSerial = true;
};

// Epilogue.  With GroupNorm statistics on (EPI_STD, f32 output feeding a GroupNorm32) every wave also emits (sum, sum of
// squares) of the values it just produced, per 16-column strip of its TM-row tile:
// gn_part[row_tile][slot][n / 16][2], slot 1 = rows that belong to the NEXT sequence when the row tile straddles a
// sequence boundary.  The GroupNorm apply kernel adds these up in a fixed order (deterministic), which removes the separate
// statistics pass over the tensor.
template <typename Epi, int FM, int FN, int TM, int TN, bool AL, typename Ops>
__device__ __forceinline__ void run_epilogue(const GemmCore& c, const typename Epi::Args& e, f32x4 (&acc)[FN][FM], const Ops& o, int step_t,
                                             int m0w, int n0w, int lane, int z) {
  const int fr = lane & 15, fg = lane >> 4;
  if constexpr (Epi::kId == 3) {  // GEGLU: fragments come in (value, gate) pairs
    static_assert(FN % 2 == 0, "GEGLU epilogue needs an even number of 16-column strips per wave");
#pragma unroll
    for (int i = 0; i < FN; i += 2) {
      const int n = n0w + i * 16 + fg * 4;
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = m0w + j * 16 + fr;
        Epi::apply(e, acc[i][j], o.bv[i], make_float4(0.f, 0.f, 0.f, 0.f));
        Epi::apply(e, acc[i + 1][j], o.bv[i + 1], make_float4(0.f, 0.f, 0.f, 0.f));
        if (m < c.M && n < c.N) Epi::store_pair(e, m, n, acc[i][j], acc[i + 1][j]);
      }
    }
    (void)step_t; (void)z;
    return;
  }
  constexpr bool BIG = FM * FN > 8;
  bool stats = false;
  if constexpr (Epi::kId == 0) {
    if constexpr (Epi::kStats < 0) stats = e.gn_part != nullptr && !Epi::slab(e);
    else stats = Epi::kStats > 0;
  }
  const int rt = m0w / TM;  // row-tile index (m0w is a multiple of TM, a power of two)
  int next_start = 0x7fffffff, b_first = 0;
  int end0 = 0x7fffffff, end1 = 0x7fffffff;  // padded batches: first row PAST the valid rows of the two sequences this tile can touch
  if constexpr (Epi::kId == 0) {
    if (stats) {
      unsigned r_;
      b_first = (int)fdiv((unsigned)m0w, e.gn_seq, r_);
      next_start = (b_first + 1) * (int)e.gn_seq.d;  // first row of the next sequence
      if (e.gn_vperiod > 0) {
        end0 = b_first * (int)e.gn_seq.d + e.gn_vlen[b_first % e.gn_vperiod];
        end1 = next_start + e.gn_vlen[(b_first + 1) % e.gn_vperiod];
      }
    }
  }
  // phase 2: arithmetic and GroupNorm partial statistics, registers only
  float s0[FN], q0[FN], s1[FN], q1[FN];
  // big wave tiles read the skip here, strip by strip.  With the strip's own four quads as the only loads in flight the 256 x 256 tile's
  // epilogue is a chain of four dependent HBM round trips per wave (~40 us per tile at the pre-pass shapes, as long as its 16-k-tile main
  // loop).  Requesting the NEXT strip's quads before this strip is worked on (AHEAD, round 5) needs 16 more live registers in a 128-VGPR
  // wave: hipcc spills 40 bytes per lane and the long-form reading workload ran 2.3 % slower (profiles/r05_ab_flash_ks4_and_epilogue_prefetch.txt)
  // - the knob stays off.
#if defined(TT_EPI_PREFETCH)   // A/B knob (build.py --variant pf -DTT_EPI_PREFETCH), default OFF: measured slower, see below
  constexpr bool AHEAD = true;
#else
  constexpr bool AHEAD = false;
#endif
  constexpr bool LATE = Epi::kId == 0 && FM * FN > 8;
  float4 rq_cur[LATE ? FM : 1], rq_nxt[LATE ? FM : 1];
  auto fetch_skip = [&](float4 (&dst)[LATE ? FM : 1], int i) {
    if constexpr (LATE) {
      const int n = n0w + i * 16 + fg * 4;
      const int nvalid = c.N - n >= 4 ? 4 : c.N - n;
#pragma unroll
      for (int j = 0; j < FM; ++j) dst[j] = Epi::has_res(e) ? Epi::template load_res<AL>(c, e, m0w + j * 16 + fr, n, nvalid) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch_skip(rq_cur, 0);
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0w + i * 16 + fg * 4;
    const int nvalid = c.N - n >= 4 ? 4 : c.N - n;
    s0[i] = q0[i] = s1[i] = q1[i] = 0.f;
    if constexpr (LATE) {
      if constexpr (AHEAD) {
        if (i + 1 < FN) fetch_skip(rq_nxt, i + 1);
      } else if (i > 0) {
        fetch_skip(rq_cur, i);
      }
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0w + j * 16 + fr;
      if constexpr (Epi::kId == 0) {
        float4 rq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (Epi::template late_res<FM, FN>()) {
          rq = rq_cur[j];
        } else {
          rq = o.rv[i][j];
        }
        Epi::apply(e, acc[i][j], o.bv[i], rq);
      } else {
        Epi::apply(e, acc[i][j], o.bv[i], make_float4(0.f, 0.f, 0.f, 0.f));
      }
      if (stats) {
        float sv = 0.f, qv = 0.f;
        const bool first = m < next_start;
        const bool row_ok = m < c.M && m < (first ? end0 : end1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ev = (r < nvalid && row_ok) ? acc[i][j][r] : 0.f;
          sv += ev;
          qv += ev * ev;
        }
        s0[i] += first ? sv : 0.f;
        q0[i] += first ? qv : 0.f;
        s1[i] += first ? 0.f : sv;
        q1[i] += first ? 0.f : qv;
      }
    }
    // big wave tiles (16 waves per CU, 128 VGPRs): finish one 16-column strip at a time - its stores go out right here, so the
    // accumulators die strip by strip instead of all FM * FN quads, their skip reads and their store addresses being live at once
    if constexpr (BIG) {
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = m0w + j * 16 + fr;
        if (m < c.M && n < c.N) Epi::template store<AL>(c, e, step_t, m, n, acc[i][j], nvalid, z);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LATE && AHEAD) {
#pragma unroll
        for (int j = 0; j < FM; ++j) rq_cur[j] = rq_nxt[j];
      }
    }
  }
  // phase 3: stores, back to back
  if constexpr (!BIG) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0w + i * 16 + fg * 4;
      const int nvalid = c.N - n >= 4 ? 4 : c.N - n;
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = m0w + j * 16 + fr;
        if (m < c.M && n < c.N) Epi::template store<AL>(c, e, step_t, m, n, acc[i][j], nvalid, z);
      }
    }
  }
  if constexpr (Epi::kId == 0) {
    if (stats) {
      const bool straddle = m0w + TM - 1 >= next_start;  // wave-uniform
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int n16 = (n0w + i * 16) >> 4;
        if (n0w + i * 16 < c.N) {
          const float a0 = wave_sum(s0[i]), b0 = wave_sum(q0[i]);
          float a1 = 0.f, b1 = 0.f;
          if (straddle) {
            a1 = wave_sum(s1[i]);
            b1 = wave_sum(q1[i]);
          }
          if (lane == 0) {
            float* p = e.gn_part + (((size_t)rt * 2 + 0) * e.gn_ncol16 + n16) * 2;
            *(float2*)p = make_float2(a0, b0);
            float* p1 = e.gn_part + (((size_t)rt * 2 + 1) * e.gn_ncol16 + n16) * 2;
            *(float2*)p1 = make_float2(a1, b1);
          }
        }
      }
    }
  }
#if defined(TT_WT_F32)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------------------------------
// Direct-to-LDS staging (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without passing through VGPRs or
// ds_write instructions.  A wave instruction fills 1 KiB = 8 rows x 128 B, lane-linear, so rows are
// unpadded; bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS chunk c of row r
// holds global 16-byte chunk c ^ ((r >> 1) & 7), and fragment reads apply the same involution.
// Conv padding / out-of-range rows cannot be zero-filled by a select any more: those lanes read a 16-byte
// zero page instead.
static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// HA2: second activation source: -1 run-time test of c.A2, 0 never, 1 always
template <typename T, int BM, int BN, int NW, int WM, int ST, typename Epi, bool CONV, bool AL, int HA2>
__global__ __launch_bounds__(NW * 64) void gemm_glds_kernel(const GemmDev<typename Epi::Args> g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BK = 64;
  constexpr int WGN = NW / WM;        // waves along N; WM waves along M
  constexpr int TM = BM / WM, TN = BN / WGN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;  // 1-KiB pieces (8 rows) per wave per stage
  static_assert(PA >= 1 && PW >= 1, "tile too small for this many waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As = (T*)smem_raw;             // [ST][BM][64]
  T* Ws = As + ST * BM * BK;        // [ST][BN][64]
  const GemmCore& c = g.c;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  // XCD-aware tile order.  Hardware deals workgroup i to XCD i % 8, each with a private 4 MiB L2, and everything that is
  // not in the LOCAL L2 arrives over the fabric at HBM-like bandwidth (~6.5 TB/s for the whole chip, Infinity-Cache hits
  // included: scripts/kbench.py bw).  So the tile grid is cut into row bands and every XCD owns a contiguous run of
  // (band, column, row-in-band)-ordered tiles, i.e. a rectangle of about (gx / bands) x (8 gy / ...) tiles: it pulls
  // A / bands + W * bands / 8 over the fabric instead of all of A (one band, the decode shapes where A is tiny) or all
  // of W (8 bands).  gemm_launch picks the band count to minimise that sum.  The grid is one-dimensional
  // (gx * gy workgroups, z = split-K slab) and every division is a multiply-high by a host-computed reciprocal.
  unsigned bx, by;
  {
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
  const int z = blockIdx.z;
  // split-K slab z covers k-tiles [kt_begin, kt_end): nk_total / splitk each, the first nk_total % splitk slabs one more.
  const int kt_begin = z * c.sk_quot + min(z, c.sk_rem);
  const int kt_end = kt_begin + c.sk_quot + (z < c.sk_rem ? 1 : 0);
  const T* A = (const T*)c.A;
  const T* W = (const T*)c.W;
  const T* zero = (const T*)g_zero_page;
  const T* A2 = nullptr;
  if constexpr (!CONV && HA2 != 0) {
    if (HA2 > 0 || c.A2) A2 = (const T*)c.A2 + (c.a2_slot ? (size_t)(*c.a2_slot) * c.a2_slot_stride : 0);
  }

  // per-piece lane geometry: this lane fills LDS chunk lc of row (piece * 8 + lr) with global chunk lc ^ swz(row)
  const int lr = lane >> 3, lc = lane & 7;
  int a_b[PA], a_s[PA], a_src[PA];
  bool a_ok[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = (wave + NW * p) * 8 + lr;
    a_src[p] = (lc ^ ((row >> 1) & 7)) * 8;
    const int m = m0 + row;
    a_ok[p] = m < c.M;
    if (CONV) {
      unsigned s_;
      a_b[p] = (int)fdiv((unsigned)m, c.seq, s_);
      a_s[p] = (int)s_;
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
    w_ptr[p] = W + (size_t)(n < c.N ? n : c.N - 1) * c.ldw + (lc ^ ((row >> 1) & 7)) * 8;
  }

  // k-tile cursor of the NEXT tile to request: (it, tap, k-tile within the tap).  Advanced incrementally (no division
  // in the loop); it stops at the last tile, which is then re-requested into a ring slot that is never read again.
  const int last = kt_end - 1;
  int it = min(kt_begin, last), it_tap = 0, it_kin = it;
  if (CONV && it > 0) {  // conv + split-K only: one real division, off the common path
    it_tap = it / c.cin_tiles;
    it_kin = it - it_tap * c.cin_tiles;
  }
  auto issue = [&](int buf) {
    const int kin = it_kin * BK;
    const int shift = (it_tap - c.taps_half) * c.dil;
    T* as = As + buf * BM * BK;
    T* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const T* src;
      if (CONV) {
        const int s2 = a_s[p] + shift;
        const bool ok = a_ok[p] && s2 >= 0 && s2 < c.seq_len;
        src = ok ? A + ((size_t)a_b[p] * c.seq_len + s2) * c.lda + kin + a_src[p] : zero;
      } else if (HA2 != 0) {
        const bool second = A2 != nullptr && it >= c.a2_tile;  // block-uniform: k-tiles never straddle k_split (multiple of 64)
        src = second ? A2 + (size_t)a_s[p] * c.lda2 + (kin - c.a2_tile * BK) + a_src[p] : A + (size_t)a_s[p] * c.lda + kin + a_src[p];
      } else {
        src = A + (size_t)a_s[p] * c.lda + kin + a_src[p];
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(as + (wave + NW * p) * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_ptr[p] + (size_t)it * BK), (lds_void_t*)(ws + (wave + NW * p) * 8 * BK), 16, 0, 0);
    if (it < last) {
      ++it;
      ++it_kin;
      if (CONV && it_kin == c.cin_tiles) {
        it_kin = 0;
        ++it_tap;
      }
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  typename Epi::template Ops<FM, FN> eo;
  // serial split-K (EpiSerial): running value res + bias + P0 + P1 + ..., folded every `ser_per` k-tiles
  f32x4 tser[Epi::kSerial ? FN : 1][Epi::kSerial ? FM : 1];
  int ser_per = 0, ser_left = 0;
  auto ser_fold = [&]() {
    if constexpr (Epi::kSerial) {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          tser[i][j][0] += acc[i][j][0]; tser[i][j][1] += acc[i][j][1]; tser[i][j][2] += acc[i][j][2]; tser[i][j][3] += acc[i][j][3];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      ser_left = ser_per;
    }
  };
  auto ser_begin = [&]() {  // after the operand fetch has been issued: x + bias, exactly the LayerNorm kernel's `t = x; t += bias`
    if constexpr (Epi::kSerial) {
      ser_per = (kt_end - kt_begin) / g.e.splitk;
      ser_left = ser_per;
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          tser[i][j] = f32x4{eo.rv[i][j].x + eo.bv[i].x, eo.rv[i][j].y + eo.bv[i].y, eo.rv[i][j].z + eo.bv[i].z, eo.rv[i][j].w + eo.bv[i].w};
    }
  };

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
    // two-stage variant: every barrier drains the queue anyway, so the epilogue operands go out with the first tile
    // (inside the loop, even on the last iteration only, the request de-pipelines the loop: CLVP 0.033 -> 0.037 s)
    issue(0);
    Epi::template fetch<FM, FN, AL>(c, g.e, eo, m0 + wm * TM, n0 + wn * TN, lane);
    __syncthreads();  // (drains the LDS-DMA: hipcc emits vmcnt(0) before the barrier while a global_load_lds is pending)
    ser_begin();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (kt + 1 < kt_end) issue(cur ^ 1);
      compute(cur);
      if constexpr (Epi::kSerial) {
        if (--ser_left == 0) ser_fold();
      }
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // ST-stage ring, ST-1 tiles in flight.  One raw barrier per k-step; the wait is a COUNTED vmcnt so the
    // newer stages stay in flight across the barrier (a __syncthreads() here would drain them: vmcnt(0)).
    // Every iteration issues exactly G loads (tile index clamped; a redundant reload targets the ring slot
    // that was consumed last iteration and is never read again), which keeps the count uniform in the tail.
    constexpr int G = PA + PW;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) issue(s);
    // Bias / residual operands are requested AFTER the ring fill: memory operations retire in order, so a residual quad
    // requested first would have to land before the first k-step may start; here it only has to land before stage ST - 1
    // is consumed (the counted waits below over-wait by these few loads during the first two k-steps, nothing more).
    Epi::template fetch<FM, FN, AL>(c, g.e, eo, m0 + wm * TM, n0 + wn * TN, lane);
    if constexpr (Epi::kSerial) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the skip / bias quads (and the ring fill in front of them) have landed
      ser_begin();
    }
    int slot = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
      __builtin_amdgcn_s_barrier();
      int nslot = slot + ST - 1;
      if (nslot >= ST) nslot -= ST;
      issue(nslot);
      compute(slot);
      if constexpr (Epi::kSerial) {
        if (--ser_left == 0) ser_fold();
      }
      slot = slot + 1 == ST ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  if constexpr (Epi::kSerial) {
    typedef EpiStd<T, ACT_NONE, 0, EB_F32> EOut;  // the running value already holds skip + bias: store it, nothing else
    typename EOut::template Ops<FM, FN> none;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      none.bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < FM; ++j) none.rv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    run_epilogue<EOut, FM, FN, TM, TN, AL>(c, g.e, tser, none, 0, m0 + wm * TM, n0 + wn * TN, lane, z);
  } else {
    run_epilogue<Epi, FM, FN, TM, TN, AL>(c, g.e, acc, eo, eo.step(), m0 + wm * TM, n0 + wn * TN, lane, z);
  }
}


// ------------------------------------------------------------------------------------------------------
// 3-tap convolution with the A tile shared between the taps.  The tap-as-k-tile kernel above requests every activation
// row three times (once per tap) and, at the denoiser's shapes, sits on the per-CU L2 -> LDS bandwidth: a
// 128 x 64 tile moves 24 KB per k-step for 1 MFMA-step of work.  Here a ring stage holds, for ONE 64-channel slice, the
// (BM + 2)-row halo tile of A (rows m0 - 1 .. m0 + BM) and the three taps' W tiles: 42 KB for three k-steps of work (14 KB
// per step, -42 %), one barrier per three k-steps.  Tap t of output row r reads LDS row r + t of the same tile.  Rows whose
// neighbour lies in another sequence cannot be zero-filled at load time any more (the same LDS row is a valid neighbour for
// one output row and padding for another): the fragment registers of those rows are zeroed after the LDS read, in waves that
// contain a sequence edge (wave-uniform test).  Accumulation order is slice-major (tap inside), not tap-major as above.
// LDS image of a stage: A pieces 0 .. APIECES-1 (8 rows x 128 B each; body row r at LDS row r + 8, the two halo rows at LDS rows
// 7 and BM + 8, the rest of those two pieces reads the zero page), then W pieces tap-major.  The PIECES 1-KiB pieces are dealt
// round-robin to the NW waves: waves < PIECES % NW carry one more, so the counted vmcnt has two (wave-uniform) values.
template <typename T, int BM, int BN, int NW, int WM, int ST, typename Epi, bool AL>
__global__ __launch_bounds__(NW * 64) void gemm_conv3s_kernel(const GemmDev<typename Epi::Args> g) {
  typedef typename Vec<T>::x8 x8;
  constexpr int BK = 64, TAPS = 3;
  constexpr int WGN = NW / WM;
  constexpr int TM = BM / WM, TN = BN / WGN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int AROWS = BM + 16, APIECES = AROWS / 8, WPT = BN / 8, WPIECES = TAPS * WPT;
  constexpr int PIECES = APIECES + WPIECES;
  constexpr int PMAX = (PIECES + NW - 1) / NW, PMIN = PIECES / NW, NBIG = PIECES % NW;
  constexpr int STAGE = PIECES * 512;  // elements per ring stage
  static_assert(PMAX <= 8 && ST >= 3, "conv3s geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ring = (T*)smem_raw;
  const GemmCore& c = g.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  unsigned bx, by;
  {
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
  const T* A = (const T*)c.A;
  const T* W = (const T*)c.W;
  const T* zero = (const T*)g_zero_page;
  const int lr = lane >> 3, lc = lane & 7;
  const int cin = c.cin_tiles * BK;

  // this wave's pieces: source pointer for slice 0 and per-slice advance (0 for zero-page lanes)
  const T* src[PMAX];
  int adv[PMAX];
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int p = wave + NW * k;  // wave-uniform
    src[k] = zero;
    adv[k] = 0;
    if (p < APIECES) {
      const int R = p * 8 + lr;                     // LDS row of the A image
      const int m = m0 + R - 8;                     // activation row it mirrors
      const bool wanted = R >= 7 && R <= BM + 8;    // body + the two halo rows
      if (wanted && m >= 0 && m < c.M) {
        src[k] = A + (size_t)m * c.lda + (lc ^ ((R >> 1) & 7)) * 8;
        adv[k] = BK;
      }
    } else if (p < PIECES) {
      const int q = p - APIECES, t = q / WPT, row = (q - t * WPT) * 8 + lr;
      const int n = min(n0 + row, c.N - 1);
      src[k] = W + (size_t)n * c.ldw + (size_t)t * cin + (lc ^ ((row >> 1) & 7)) * 8;
      adv[k] = BK;
    }
  }
  const int nslice = c.cin_tiles;
  int issued = 0;
  auto issue = [&](int buf) {
    T* st = ring + (size_t)buf * STAGE;
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
      const int p = wave + NW * k;
      if (p < PIECES) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src[k], (lds_void_t*)(st + p * 512), 16, 0, 0);
        if (issued + 1 < nslice) src[k] += adv[k];  // the last slice is re-requested in the tail (uniform request count)
      }
    }
    if (issued + 1 < nslice) ++issued;
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  typename Epi::template Ops<FM, FN> eo;
  const int fr = lane & 15, fg = lane >> 4;
  // sequence edges inside this wave's rows: tap 0 of a sequence's first row and tap 2 of its last row read padding
  bool first_row[FM], last_row[FM];
  bool edge = false;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    unsigned sp;
    (void)fdiv((unsigned)(m0 + wm * TM + j * 16 + fr), c.seq, sp);
    first_row[j] = sp == 0u;
    last_row[j] = (int)sp == c.seq_len - 1;
    edge = edge || first_row[j] || last_row[j];
  }
  const bool wave_edge = __builtin_amdgcn_ballot_w64(edge) != 0ull;

  auto compute = [&](int buf, auto masked) {
    constexpr bool MASK = decltype(masked)::value;
    const T* as = ring + (size_t)buf * STAGE;
    const T* ws = as + APIECES * 512;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        x8 fa[FM], fw[FN];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int R = 8 + wm * TM + j * 16 + fr + (t - 1);
          fa[j] = *(const x8*)(as + R * BK + (((ks * 4 + fg) ^ ((R >> 1) & 7)) * 8));
          if (MASK) {
            const bool pad = (t == 0 && first_row[j]) || (t == 2 && last_row[j]);
            if (pad) {
#pragma unroll
              for (int q = 0; q < 8; ++q) fa[j][q] = (T)0.f;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const int r = wn * TN + i * 16 + fr;
          fw[i] = *(const x8*)(ws + (t * BN + r) * BK + (((ks * 4 + fg) ^ ((r >> 1) & 7)) * 8));
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
      }
    }
  };

#pragma unroll
  for (int s_ = 0; s_ < ST - 1; ++s_) issue(s_);
  Epi::template fetch<FM, FN, AL>(c, g.e, eo, m0 + wm * TM, n0 + wn * TN, lane);
  int slot = 0;
  const bool big = wave < NBIG;  // this wave requests PMAX pieces per stage
  for (int js = 0; js < nslice; ++js) {
    if (big) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PMAX) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PMIN) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + ST - 1;
    if (nslot >= ST) nslot -= ST;
    issue(nslot);
    if (wave_edge) compute(slot, std::true_type{});
    else compute(slot, std::false_type{});
    slot = slot + 1 == ST ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  run_epilogue<Epi, FM, FN, TM, TN, AL>(c, g.e, acc, eo, eo.step(), m0 + wm * TM, n0 + wn * TN, lane, 0);
}

// ---------------------------------------------------------------------------------------------- host side (per operand type)
}  // namespace tt
#include "gemm_gna.h"
#include "gemm_p8.h"
namespace tt {

extern bool g_gemm_p8;  // gemm.hip; tt_gemm_variant: 0 = the 16-wave 256 x 256 tile everywhere (A/B runs)
constexpr int kP8Smem = 2 * 4 * 128 * 64 * 2;
// the 8-wave eight-phase 256 x 256 kernel takes this launch (gemm_p8.h: plain 1 x 1, one K source, no split-K, even k-tile count, 31-bit byte offsets)
static inline bool p8_ok(const GemmCore& c, const dim3& grid) {
  return grid.z == 1 && c.sk_rem == 0 && c.sk_quot >= 2 && (c.sk_quot & 1) == 0 && c.cin_tiles == c.sk_quot && c.A2 == nullptr &&
         ((size_t)c.M * c.lda + (size_t)c.sk_quot * 64) * 2 < 0x7fffffffull && ((size_t)c.N * c.ldw + (size_t)c.sk_quot * 64) * 2 < 0x7fffffffull;
}

// TILE_32x16 / TILE_64x16 (round 6): "skinny" tiles for the weight-streaming GEMMs of a SMALL decode batch (M <= 64 rows, N >= 1024).  With
// 64 x 64 tiles a 32-row decode GEMM occupies 48 - 64 of the 256 CUs and every one of them pulls 192 - 256 KB through its ~50 GB/s
// L2 -> LDS path; 16-column tiles spread the same weight stream over 192 - 256 CUs at 96 KB each.  Same MFMA, same k order per output
// element, same split-K ranges: a row's bits do not depend on which tile computed it (asserted by the batch-independence tests).
enum Tile { TILE_64x64 = 0, TILE_128x64 = 1, TILE_128x128 = 2, TILE_256x256 = 3, TILE_32x16 = 4, TILE_64x16 = 5, TILE_COUNT = 6 };
// build knobs of the A/B runs (build.py --variant): ring depth, tile width, waves (measured: profiles/r06_ab_small_batch_decode.txt)
#ifndef TT_SKINNY_ST
#define TT_SKINNY_ST 8     // 32 x 16: 4 / 8 / 12 stages = 1.117 / 1.078 / 1.118 ms per step at 32 candidates
#endif
#ifndef TT_SKINNY_ST64
#define TT_SKINNY_ST64 4   // 64 x 16: 4 / 8 / 12 stages = 1.237 / 1.282 / 1.355 ms per step at 64 candidates (the split-K projections have 4 k-tiles per range:
#endif                     // a deeper ring only re-requests the last tile)
#ifndef TT_SKINNY_BN
#define TT_SKINNY_BN 16
#endif
#ifndef TT_SKINNY_NW
#define TT_SKINNY_NW 2
#endif
constexpr int kSkinnyStages = TT_SKINNY_ST, kSkinnyStages64 = TT_SKINNY_ST64, kSkinnyBN = TT_SKINNY_BN, kSkinnyNW = TT_SKINNY_NW, kSkinnyWM = 2;
// EPI_STD kernel variants: the generic one tests everything at run time, the others compile the unused features out
enum StdVariant { V_GEN = 0, V_NONE = 1, V_SLAB = 2, V_GELU_T = 3, V_ST_F32 = 4, V_ST_RES = 5, V_ST_A2 = 6, V_BIAS_T = 7, V_SERIAL = 8, V_COUNT = 9 };
constexpr int kNoKernel = -100;  // visit_*: this combination is not instantiated

struct GemmPlan {   // what gemm_launch (gemm.hip) decided: tile, grid, the device argument core
  GemmCore core;
  int tile;
  bool conv3s;     // shared-halo 3-tap convolution kernel
  int splitk;
  int prof_id;
  double flops, bytes;
};

template <int BM, int BN, int ST>
constexpr int smem_bytes_glds() {
  return ST * (BM + BN) * 64 * 2;
}

template <typename T, int BM, int BN, int NW, int WM, int ST, typename Epi, bool CONV, bool AL, int HA2>
struct KernelRef {
  typedef typename Epi::Args EA;
  static constexpr int smem = smem_bytes_glds<BM, BN, ST>();
  static constexpr int threads = NW * 64;
  // (kId 2: the decode-step QKV scatter never runs on this tile; kId 3: the GEGLU epilogue's 16 output columns per quadrant make 32-byte row
  //  segments - measured 9 % slower than the 16-wave tile's 64-byte ones on one box, profiles/r05_ab_gemm_eight_phase.txt - so it stays there)
  static constexpr bool kP8 = BM == 256 && BN == 256 && !CONV && AL && HA2 == 0 && !Epi::kSerial && Epi::kId != 2 && Epi::kId != 3;
  static const void* fn() { return (const void*)gemm_glds_kernel<T, BM, BN, NW, WM, ST, Epi, CONV, AL, HA2>; }
  static const void* fn_p8() {
    if constexpr (kP8) return (const void*)gemm_p8_kernel<T, Epi>;
    else return nullptr;
  }
  static void launch(const ProfScope& ps, dim3 grid, hipStream_t s, const GemmDev<EA>& d) {
    if constexpr (kP8) {
      if (g_gemm_p8 && p8_ok(d.c, grid)) {
        launch_timed(ps, gemm_p8_kernel<T, Epi>, grid, dim3(512), kP8Smem, s, d);
        return;
      }
    }
    launch_timed(ps, gemm_glds_kernel<T, BM, BN, NW, WM, ST, Epi, CONV, AL, HA2>, grid, dim3(NW * 64), smem, s, d);
  }
};

// v(KernelRef) is called for the EPI_STD instantiation (tile, variant, conv, al); kNoKernel when that one does not exist
template <typename T, int BM, int BN, int NW, int WM, int ST, typename V>
static int visit_std_tile(int variant, bool conv, bool al, V&& v) {
  typedef EpiStd<T, -1, -1, -1> EGen;                                     // everything tested at run time
  typedef EpiStd<T, ACT_NONE, 0, -1> ENone;                               // no activation / statistics, run-time outputs
  typedef EpiStd<T, ACT_NONE, 0, EB_SLAB> ESlab;                          // split-K partial sums (decode projections)
  typedef EpiStd<T, ACT_GELU_TANH, 0, EB_BIAS | EB_T> EGeluT;             // GPT-2 c_fc
  typedef EpiStd<T, ACT_NONE, 0, EB_BIAS | EB_T> EBiasT;                  // plain Linear / conv feeding the next GEMM
  typedef EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_F32> EStF32;                // denoiser 1x1 in front of a GroupNorm
  typedef EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_RES | EB_F32> EStRes;       // denoiser conv / attention projection + skip
  switch (variant) {
    case V_GEN:
      if (conv) return al ? v(KernelRef<T, BM, BN, NW, WM, ST, EGen, true, true, -1>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, EGen, true, false, -1>{});
      return al ? v(KernelRef<T, BM, BN, NW, WM, ST, EGen, false, true, -1>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, EGen, false, false, -1>{});
    case V_NONE:
      if (!al) return kNoKernel;
      return conv ? v(KernelRef<T, BM, BN, NW, WM, ST, ENone, true, true, 0>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, ENone, false, true, 0>{});
    case V_SLAB:
      if (!al || conv) return kNoKernel;
      return v(KernelRef<T, BM, BN, NW, WM, ST, ESlab, false, true, 0>{});
    case V_GELU_T:
      if (!al || conv) return kNoKernel;
      return v(KernelRef<T, BM, BN, NW, WM, ST, EGeluT, false, true, 0>{});
    case V_ST_F32:
      if (!al) return kNoKernel;
      return conv ? v(KernelRef<T, BM, BN, NW, WM, ST, EStF32, true, true, 0>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, EStF32, false, true, 0>{});
    case V_ST_RES:
      if (!al) return kNoKernel;
      return conv ? v(KernelRef<T, BM, BN, NW, WM, ST, EStRes, true, true, 0>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, EStRes, false, true, 0>{});
    case V_ST_A2:
      if (!al || conv) return kNoKernel;
      return v(KernelRef<T, BM, BN, NW, WM, ST, EStF32, false, true, 1>{});
    case V_BIAS_T:
      if (!al) return kNoKernel;
      return conv ? v(KernelRef<T, BM, BN, NW, WM, ST, EBiasT, true, true, 0>{}) : v(KernelRef<T, BM, BN, NW, WM, ST, EBiasT, false, true, 0>{});
    case V_SERIAL:
      if constexpr ((BM / WM / 16) * (BN / (NW / WM) / 16) > 8) return kNoKernel;  // needs the skip quads in registers next to two accumulator sets
      else {
        if (!al || conv) return kNoKernel;
        return v(KernelRef<T, BM, BN, NW, WM, ST, EpiSerial<T>, false, true, 0>{});
      }
  }
  return kNoKernel;
}
// the skinny tiles exist for the aligned 1 x 1 forms of a decode step only (slabs, bias + gelu + T, bias + T, run-time outputs, generic)
template <typename T, int BM, int BN, typename V>
static int visit_skinny(int variant, bool conv, bool al, V&& v) {
  if (conv || !al) return kNoKernel;
  constexpr int NW = kSkinnyNW, WM = kSkinnyWM, ST = BM == 32 ? kSkinnyStages : kSkinnyStages64;
  switch (variant) {
    case V_GEN: return v(KernelRef<T, BM, BN, NW, WM, ST, EpiStd<T, -1, 0, -1>, false, true, 0>{});
    case V_NONE: return v(KernelRef<T, BM, BN, NW, WM, ST, EpiStd<T, ACT_NONE, 0, -1>, false, true, 0>{});
    case V_SLAB: return v(KernelRef<T, BM, BN, NW, WM, ST, EpiStd<T, ACT_NONE, 0, EB_SLAB>, false, true, 0>{});
    case V_GELU_T: return v(KernelRef<T, BM, BN, NW, WM, ST, EpiStd<T, ACT_GELU_TANH, 0, EB_BIAS | EB_T>, false, true, 0>{});
    case V_BIAS_T: return v(KernelRef<T, BM, BN, NW, WM, ST, EpiStd<T, ACT_NONE, 0, EB_BIAS | EB_T>, false, true, 0>{});
  }
  return kNoKernel;
}
template <typename T, typename V>
static int visit_std(int tile, int variant, bool conv, bool al, V&& v) {
  switch (tile) {
    case TILE_32x16: return visit_skinny<T, 32, kSkinnyBN>(variant, conv, al, v);
    case TILE_64x16: return visit_skinny<T, 64, kSkinnyBN>(variant, conv, al, v);
    case TILE_256x256: return visit_std_tile<T, 256, 256, 16, 4, 2>(variant, conv, al, v);   // 4 x 4 waves of 64 x 64, two 64 KB stages
    case TILE_128x128: return visit_std_tile<T, 128, 128, 8, 2, 2>(variant, conv, al, v);
    case TILE_128x64: return visit_std_tile<T, 128, 64, 8, 4, 4>(variant, conv, al, v);   // 4 x 2 waves of 32 x 32: 1 LDS fragment read per MFMA (2 x 4 of 64 x 16: 1.25)
    default: return visit_std_tile<T, 64, 64, 4, 2, 4>(variant, conv, al, v);
  }
}
template <typename T, typename Epi, typename V>
static int visit_qkv(int tile, V&& v) {
  if (tile == TILE_32x16 || tile == TILE_64x16) {  // decode-step QKV scatter only
    if constexpr (Epi::kId == 2) {
      if (tile == TILE_32x16) return v(KernelRef<T, 32, kSkinnyBN, kSkinnyNW, kSkinnyWM, kSkinnyStages, Epi, false, true, 0>{});
      return v(KernelRef<T, 64, kSkinnyBN, kSkinnyNW, kSkinnyWM, kSkinnyStages64, Epi, false, true, 0>{});
    } else {
      return kNoKernel;
    }
  }
  switch (tile) {
    case TILE_256x256: return v(KernelRef<T, 256, 256, 16, 4, 2, Epi, false, true, 0>{});
    case TILE_128x128: return v(KernelRef<T, 128, 128, 8, 2, 2, Epi, false, true, 0>{});
    // (round 4: a 96 x 128 tile - 456 tiles = one round at two workgroups per CU instead of 336 tiles of 128 x 128 = 1.3 rounds - changed
    //  nothing on the denoiser's QKV GEMM, 1.4795 / 1.4841 vs 1.4994 / 1.4893 ms per iteration: profiles/r04_ab_geometry.txt)
    case TILE_128x64: return v(KernelRef<T, 128, 64, 8, 4, 4, Epi, false, true, 0>{});
    default: return v(KernelRef<T, 64, 64, 4, 2, 4, Epi, false, true, 0>{});
  }
}

static inline EpiStdArgs make_epi_std(const GemmArgs& a) {
  EpiStdArgs e;
  memset(&e, 0, sizeof(e));
  e.bias = a.bias; e.res = a.res; e.out_f32 = a.out_f32; e.out_t = a.out_t; e.gn_part = a.gn_part;
  e.ldres = a.ldres; e.ldo32 = a.ldo32; e.ldot = a.ldot; e.act = a.act; e.slope = a.slope; e.splitk = a.splitk;
  e.gn_ncol16 = a.gn_ncol16;
  e.act_t = a.act_t; e.slope_t = a.slope_t;
  e.gn_seq = make_fastdiv(a.gn_seq > 0 ? a.gn_seq : 1);
  e.gn_vperiod = a.gn_part ? a.gn_vperiod : 0;
  for (int i = 0; i < 32; ++i) e.gn_vlen[i] = a.gn_vlen[i];
  return e;
}

// shared-halo 3-tap convolution (gemm_conv3s_kernel): 128 x 64 tile, 8 waves as 4 x 2, 3-stage ring of 42 KB stages
constexpr int kConv3sStages = 3;
constexpr int kConv3sSmem = kConv3sStages * ((128 + 16) / 8 + 3 * 64 / 8) * 1024;
template <typename T, typename Epi>
static const void* conv3s_fn() { return (const void*)gemm_conv3s_kernel<T, 128, 64, 8, 4, kConv3sStages, Epi, true>; }

template <typename T>
int gemm_launch_typed(int epi, const GemmArgs& a, const GemmPlan& plan, hipStream_t stream) {
  const dim3 grid(plan.core.gx * plan.core.gy, 1, plan.splitk);
  ProfScope ps(plan.prof_id, stream, plan.flops, plan.bytes, true);
  int rc = kNoKernel;
  if (epi == EPI_STD && plan.conv3s) {  // (gemm.hip decided: aligned, statistics epilogue, bias + f32 output (+ skip), 128x64 tile)
    GemmDev<EpiStdArgs> d;
    d.c = plan.core;
    d.e = make_epi_std(a);
    if (a.res) launch_timed(ps, gemm_conv3s_kernel<T, 128, 64, 8, 4, kConv3sStages, EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_RES | EB_F32>, true>, grid, dim3(512), kConv3sSmem, stream, d);
    else launch_timed(ps, gemm_conv3s_kernel<T, 128, 64, 8, 4, kConv3sStages, EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_F32>, true>, grid, dim3(512), kConv3sSmem, stream, d);
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (epi == EPI_STD) {
    GemmDev<EpiStdArgs> d;
    d.c = plan.core;
    d.e = make_epi_std(a);
    // aligned fast path: whole-quad operand fetches and stores with no per-element fallback code in the kernel
    const bool al = (a.N & 3) == 0 && a.N >= 4 && (!a.bias || ((size_t)a.bias & 15) == 0) &&
                    (!a.res || (((size_t)a.res & 15) == 0 && (a.ldres & 3) == 0)) &&
                    (!a.out_f32 || (((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0)) && (!a.out_t || (((size_t)a.out_t & 7) == 0 && (a.ldot & 3) == 0));
    const bool conv = a.taps > 1, stats = a.gn_part != nullptr, a2 = a.A2 != nullptr;
    int variant = V_GEN;
    if (a.serial_k > 1) {
      variant = V_SERIAL;
      d.e.splitk = a.serial_k;  // K ranges folded inside the launch
    } else if (al) {
      const int mode = a.splitk > 1 ? EB_SLAB : ((a.bias ? EB_BIAS : 0) | (a.res ? EB_RES : 0) | (a.out_f32 ? EB_F32 : 0) | (a.out_t ? EB_T : 0));
      if (a.act == ACT_NONE) {
        if (stats) {
          if (!a2 && mode == (EB_BIAS | EB_F32)) variant = V_ST_F32;
          else if (!a2 && mode == (EB_BIAS | EB_RES | EB_F32)) variant = V_ST_RES;
          else if (a2 && !conv && mode == (EB_BIAS | EB_F32)) variant = V_ST_A2;
        } else if (!a2) {
          variant = (mode == EB_SLAB && !conv) ? V_SLAB : (mode == (EB_BIAS | EB_T) && !a.act_t) ? V_BIAS_T : V_NONE;
        }
      } else if (a.act == ACT_GELU_TANH && !a2 && !stats && !conv && mode == (EB_BIAS | EB_T) && !a.act_t) {
        variant = V_GELU_T;
      }
    }
    auto go = [&](auto kr) -> int {
      decltype(kr)::launch(ps, grid, stream, d);
      return 0;
    };
    rc = visit_std<T>(plan.tile, variant, conv, al, go);
    if (rc == kNoKernel && variant != V_SERIAL) rc = visit_std<T>(plan.tile, V_GEN, conv, al, go);
  } else if (epi == EPI_QKV_HEADS) {
    GemmDev<EpiQkvHeadsArgs> d;
    d.c = plan.core;
    memset(&d.e, 0, sizeof(d.e));
    d.e.bias = a.bias; d.e.q = a.q; d.e.k = a.k; d.e.v = a.v; d.e.vt = a.vt; d.e.heads = a.heads; d.e.seq_pad = a.seq_pad; d.e.q_scale = a.q_scale;
    d.e.dmodel = make_fastdiv(a.dmodel);
    rc = visit_qkv<T, EpiQkvHeads<T>>(plan.tile, [&](auto kr) -> int {
      decltype(kr)::launch(ps, grid, stream, d);
      return 0;
    });
  } else if (epi == EPI_QKV_DECODE) {
    GemmDev<EpiQkvDecodeArgs> d;
    d.c = plan.core;
    memset(&d.e, 0, sizeof(d.e));
    d.e.bias = a.bias; d.e.step = a.step; d.e.qbuf = a.qbuf; d.e.kc = a.kc; d.e.vc = a.vc; d.e.heads = a.heads; d.e.tmax = a.tmax; d.e.dmodel_i = a.dmodel;
    d.e.q_scale = a.q_scale;
    d.e.dmodel = make_fastdiv(a.dmodel);
    rc = visit_qkv<T, EpiQkvDecode<T>>(plan.tile, [&](auto kr) -> int {
      decltype(kr)::launch(ps, grid, stream, d);
      return 0;
    });
  } else if (epi == EPI_GEGLU) {
    GemmDev<EpiGegluArgs> d;
    d.c = plan.core;
    d.e.bias = a.bias; d.e.out_t = a.out_t; d.e.ldot = a.ldot;
    rc = visit_qkv<T, EpiGeglu<T>>(plan.tile, [&](auto kr) -> int {
      decltype(kr)::launch(ps, grid, stream, d);
      return 0;
    });
  }
  if (rc == kNoKernel) {
    set_error("gemm: no kernel for epilogue %d tile %d", epi, plan.tile);
    return -1;
  }
  TT_CHECK_HIP(hipGetLastError());
  return rc;
}

// GroupNorm-apply on the A path (gemm_gna.h): 32 x 256 tile, 8 waves of 32 x 32 (two per SIMD: one wave's SiLU transcendentals run under the
// other's MFMA / LDS waits), both operands two k-tiles ahead in registers.  220 workgroups at the denoiser's 1740 rows x 1024 columns, the
// apply redone by 4 column tiles.  In-situ A/Bs (profiles/r04_ab_fused_groupnorm.txt; sampler iteration, stand-alone apply = 100 %):
// 64 x 128 / 4 waves +4 %, 64 x 128 / 8 waves -1 % (box-dependent), 32 x 256 / 8 waves -2.2 .. -2.7 %; prefetch 2 beats 1 (far) and 4, 8.
// The TT_GNA_* macros are the knobs of those A/B builds (build.py --variant).
#ifndef TT_GNA_NW
#define TT_GNA_NW 8
#endif
#ifndef TT_GNA_PF
#define TT_GNA_PF 2
#endif
#ifndef TT_GNA_BM
#define TT_GNA_BM 32
#define TT_GNA_BN 256
#define TT_GNA_WM 1
#endif
constexpr int kGnaBM = TT_GNA_BM, kGnaBN = TT_GNA_BN, kGnaWM = TT_GNA_WM, kGnaST = TT_GNA_PF, kGnaNW = TT_GNA_NW;  // (kGnaST: prefetch depth in k-tiles)
constexpr int kGnaSmem = 0;
template <typename T, typename Epi, bool SS, bool SILU>
static const void* gna_fn() { return (const void*)gemm_gna_kernel<T, kGnaBM, kGnaBN, kGnaNW, kGnaWM, kGnaST, Epi, SS, SILU>; }

template <typename T>
int gemm_gna_launch_typed(const GemmArgs& a, const GemmPlan& plan, const GnaArgs& n, hipStream_t stream) {
  const dim3 grid(plan.core.gx * plan.core.gy, 1, 1);
  ProfScope ps(plan.prof_id, stream, plan.flops, plan.bytes, true);
  typedef EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_F32> EStF32;
  GemmGnaDev<EpiStdArgs> d;
  d.c = plan.core;
  d.e = make_epi_std(a);
  d.n = n;
  if (n.act == ACT_SILU && !n.ss) launch_timed(ps, gemm_gna_kernel<T, kGnaBM, kGnaBN, kGnaNW, kGnaWM, kGnaST, EStF32, false, true>, grid, dim3(kGnaNW * 64), kGnaSmem, stream, d);
  else {
    set_error("gemm_gna: no kernel for act %d scale_shift %d", n.act, n.ss != nullptr);
    return -1;
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// dynamic-LDS attribute of every instantiation this type can launch
template <typename T>
int gemm_init_typed() {
  int bad = 0;
  auto setattr = [&](auto kr) -> int {
    if (hipFuncSetAttribute(decltype(kr)::fn(), hipFuncAttributeMaxDynamicSharedMemorySize, decltype(kr)::smem) != hipSuccess) ++bad;
    if (decltype(kr)::fn_p8() && hipFuncSetAttribute(decltype(kr)::fn_p8(), hipFuncAttributeMaxDynamicSharedMemorySize, kP8Smem) != hipSuccess) ++bad;
    return 0;
  };
  for (int tile = 0; tile < TILE_COUNT; ++tile) {
    for (int variant = 0; variant < V_COUNT; ++variant)
      for (int conv = 0; conv < 2; ++conv)
        for (int al = 0; al < 2; ++al) (void)visit_std<T>(tile, variant, conv != 0, al != 0, setattr);
    (void)visit_qkv<T, EpiQkvHeads<T>>(tile, setattr);
    (void)visit_qkv<T, EpiQkvDecode<T>>(tile, setattr);
    (void)visit_qkv<T, EpiGeglu<T>>(tile, setattr);
  }
  if (hipFuncSetAttribute(conv3s_fn<T, EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_RES | EB_F32>>(), hipFuncAttributeMaxDynamicSharedMemorySize, kConv3sSmem) != hipSuccess) ++bad;
  if (hipFuncSetAttribute(conv3s_fn<T, EpiStd<T, ACT_NONE, 1, EB_BIAS | EB_F32>>(), hipFuncAttributeMaxDynamicSharedMemorySize, kConv3sSmem) != hipSuccess) ++bad;
  if (bad) {
    set_error("gemm: hipFuncSetAttribute failed for %d kernel(s): %s", bad, hipGetErrorString(hipGetLastError()));
    return -2;
  }
  return 0;
}

}  // namespace tt
