// Attention kernels for gfx950, head_dim = 64.
//
// flash_attention: one wave owns NQ blocks of 16 queries and walks the keys 32 at a time with an
// online softmax.  Both products run "swapped" so the softmax axis is lane-local:
//   S^T[key][q] = K[key][:] . Q[q][:]        (MFMA A = K rows, B = Q rows: both 16-B row reads)
//   O^T[d][q]   = V^T[d][key] . P^T[key][q]  (MFMA A = V^T rows, B = P straight from the S^T registers)
// A lane therefore holds one query column (l & 15) in every accumulator: running max / sum and
// the rescale factor are per-lane scalars and the row reductions are two xor-shuffles (16, 32).
// The S^T -> P^T hand-off needs no LDS: the 32 keys of a tile are fed to the second MFMA in the
// order the first one left them in registers, and V^T is read with the same permutation.
// K/V for one (batch, head) is at most a few hundred KB, i.e. L2 resident, so nothing is staged
// through LDS (the only LDS use is the 129-entry relative-position table of DiffusionTts).
//
// decode_attention: one wave per (sequence, head), one new query against [shared prefix | own
// generated keys].  HBM-bound on the per-sequence cache: keys are stored in 16-byte dim-chunks
// that are key-major so the lane-per-key dot product issues fully coalesced 1-KiB loads; values are
// row-major and read 8 whole rows (1 KiB) per wave instruction.
#include "ops.h"

namespace tt {

// SPLIT = false: each wave owns its own NQ x 16 queries and walks all keys.
// SPLIT = true : the 4 waves of a block share ONE block of queries and take every 4th key tile each; partial
//                (max, sum, O) states are merged through LDS.  4x the resident waves for the same work, which is
//                what hides the per-tile dependency chain (MFMA -> softmax -> MFMA) when batch*heads is small.
template <typename T, int NQ, bool SPLIT>
__global__ __launch_bounds__(256, 2) void flash_kernel(FlashArgs a) {  // 2 waves per SIMD: <= 256 VGPRs (NQ = 4 took 308 => one workgroup per CU)
  typedef typename Vec<T>::x8 x8;
  typedef typename Vec<T>::x4 x4;
  __shared__ float rp[132];
  // XCD-aware block order: the dispatcher deals workgroups round-robin over the 8 XCDs (private L2 each), which
  // would spread the query blocks of one (batch, head) over all eight L2s and fetch its K / V eight times
  // (PMC: FETCH_SIZE 4.4x the algorithmic bytes).  Remap so every XCD owns a contiguous run of (head, query block).
  int bx = blockIdx.x, bh = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int lin = bh * gx + bx, xcd = lin & 7, slot = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int lin2 = xcd * per + min(xcd, rem) + slot;
    bh = lin2 / gx;
    bx = lin2 - bh * gx;
  }
  const int h = bh % a.heads, b = bh / a.heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ns = a.n;                                             // row stride of the operands
  const int n = a.nv_period > 0 ? a.nv[b % a.nv_period] : a.n;    // valid keys / queries of this batch row (padded batches)
  if (a.relpos) {
    if (threadIdx.x < 129) rp[threadIdx.x] = a.relpos[h * 129 + threadIdx.x];
    __syncthreads();
  }
  const int qbase = (SPLIT ? bx : bx * 4 + wave) * 16 * NQ;
  if (qbase >= n) return;
  const T* Q = (const T*)a.q + (size_t)bh * ns * 64;
  const T* K = (const T*)a.k + (size_t)bh * ns * 64;
  const T* VT = (const T*)a.vt + (size_t)bh * 64 * a.n_pad;

  x8 qf[NQ][2];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    const int qr = min(qbase + iq * 16 + fr, n - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[iq][ks] = *(const x8*)(Q + (size_t)qr * 64 + ks * 32 + fg * 8);
  }
  float m_run[NQ], l_run[NQ];
  f32x4 acc[NQ][4];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    m_run[iq] = -1e30f;
    l_run[iq] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) acc[iq][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int q_last = min(qbase + 16 * NQ, n) - 1;
  const int kend = a.causal ? q_last + 1 : n;

  // K / V^T fragments of tile t+1 are fetched into a second register set while tile t is processed:
  // a wave's loop body is shorter than an L2 round trip, so without this every tile pays the latency.
  auto load_kv = [&](x8 (&kf)[2][2], x8 (&vf)[4], int key0) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int kr = min(key0 + kb * 16 + fr, n - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[kb][ks] = *(const x8*)(K + (size_t)kr * 64 + ks * 32 + fg * 8);
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const T* vrow = VT + (size_t)(blk * 16 + fr) * a.n_pad + key0 + fg * 4;
      const x4 lo = *(const x4*)vrow;
      const x4 hi = *(const x4*)(vrow + 16);
      x8 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      vf[blk] = v;
    }
  };
  x8 kfA[2][2], vfA[4], kfB[2][2], vfB[4];

  // The loop body is VALU-bound (PMC: ~24 VALU instructions per MFMA before this diet), so everything that is
  // wave-uniform is decided once per tile: tiles whose keys are all >= 64 positions away from every query of the
  // block take a constant relative-position bias, masks are only evaluated on the last / diagonal tile, the
  // accumulator rescale is skipped when no row maximum moved, the row sums stay lane-partial until the end, and
  // exp() is a bare v_exp_f32 (exp2) with log2(e) folded into one FMA.
  constexpr float LOG2E = 1.4426950408889634f;
  auto process = [&](const x8 (&kf)[2][2], const x8 (&vf)[4], int key0) {
    const bool tail = key0 + 32 > n;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      const int q0 = qbase + iq * 16;
      const int qi = q0 + fr;
      float s[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
        st = mfma16(kf[kb][0], qf[iq][0], st);
        st = mfma16(kf[kb][1], qf[iq][1], st);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kb][r] = st[r];
      }
      if (a.relpos) {
        if (key0 - (q0 + 15) >= 64) {          // every key is >= 64 after every query: bucket saturated
          const float bconst = rp[128];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] += bconst;
        } else if (q0 - (key0 + 31) >= 64) {   // every key is >= 64 before every query
          const float bconst = rp[0];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] += bconst;
        } else {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int d = key0 + kb * 16 + fg * 4 + r - qi;
              d = d < -64 ? -64 : (d > 64 ? 64 : d);
              s[kb][r] += rp[d + 64];
            }
        }
      }
      if (tail || (a.causal && key0 + 31 > q0)) {  // masks only on the last key tile / the causal diagonal
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kb * 16 + fg * 4 + r;
            if (key >= n || (a.causal && key > qi)) s[kb][r] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                       fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      mx = max_xor16(mx);
      mx = max_xor32(mx);
      const float m_new = fmaxf(m_run[iq], mx);
      if (__any(m_new > m_run[iq])) {  // some row maximum moved: rescale the running state (exact when skipped)
        const float alpha = __builtin_amdgcn_exp2f((m_run[iq] - m_new) * LOG2E);
        l_run[iq] *= alpha;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          acc[iq][blk][0] *= alpha; acc[iq][blk][1] *= alpha; acc[iq][blk][2] *= alpha; acc[iq][blk][3] *= alpha;
        }
        m_run[iq] = m_new;
      }
      const float mc = m_new * LOG2E;
      float p[2][4];
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, -mc));
          psum += p[kb][r];
        }
      l_run[iq] += psum;  // lane-partial: reduced over the four key groups once, after the loop
      x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (T)p[0][r];
        pf[4 + r] = (T)p[1][r];
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[iq][blk] = mfma16(vf[blk], pf, acc[iq][blk]);
    }
  };
  // prefetches are unconditional (clamped to the last tile): straight-line body, counted waits
  const int ntile = (kend + 31) / 32;
  constexpr int TSTEP = SPLIT ? 4 : 1;
  int t = SPLIT ? wave : 0;
  if (t < ntile) {
    load_kv(kfA, vfA, 32 * t);
    while (true) {
      load_kv(kfB, vfB, 32 * min(t + TSTEP, ntile - 1));
      process(kfA, vfA, 32 * t);
      t += TSTEP;
      if (t >= ntile) break;
      load_kv(kfA, vfA, 32 * min(t + TSTEP, ntile - 1));
      process(kfB, vfB, 32 * t);
      t += TSTEP;
      if (t >= ntile) break;
    }
  }
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    l_run[iq] = add_xor16(l_run[iq]);
    l_run[iq] = add_xor32(l_run[iq]);
  }
  if (SPLIT) {
    // merge the four waves' partial (max, sum, O) states, one query block at a time, into wave 0
    __shared__ float mbuf[4][16], lbuf[4][16];
    __shared__ float obuf[4][64][17];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      if (iq > 0) __syncthreads();  // wave 0 finished reading the previous query block
      if (fg == 0) {
        mbuf[wave][fr] = m_run[iq];
        lbuf[wave][fr] = l_run[iq];
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) obuf[wave][lane][blk * 4 + r] = acc[iq][blk][r];
      __syncthreads();
      if (wave == 0) {
        float mw[4], M = -1e30f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          mw[w] = mbuf[w][fr];
          M = fmaxf(M, mw[w]);
        }
        float L = 0.f;
        float o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float f = __expf(mw[w] - M);
          L += f * lbuf[w][fr];
#pragma unroll
          for (int j = 0; j < 16; ++j) o[j] += f * obuf[w][lane][j];
        }
        l_run[iq] = L;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[iq][blk][r] = o[blk * 4 + r];
      }
    }
    if (wave != 0) return;
  }
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    const int qi = qbase + iq * 16 + fr;
    if (qi < n) {
      const float inv = 1.0f / l_run[iq];
      T* o = (T*)a.out + ((size_t)b * ns + qi) * a.ldo + h * 64 + fg * 4;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
        *(x4*)(o + blk * 16) = pack4<T>(acc[iq][blk][0] * inv, acc[iq][blk][1] * inv, acc[iq][blk][2] * inv, acc[iq][blk][3] * inv);
    }
  }
}

// ------------------------------------------------------------------------------- LDS-staged flash attention
// Same arithmetic as flash_kernel (swapped QK^T / PV MFMAs, online softmax per 32 keys), different data movement: the four
// waves of a block own DIFFERENT queries (16 * NQ each) and SHARE every 64-key K / V^T tile, which goes global -> LDS
// directly (global_load_lds_dwordx4, 16 KiB per tile, XOR-swizzled on the source side like the GEMM tiles) through a 3-stage
// ring with counted vmcnt, one raw barrier per tile.  Against the register-prefetch kernel this divides the L2 -> CU traffic
// by 4 (a K / V byte is fetched once per block, not once per wave), puts two more tiles in flight per wave without spending
// VGPRs on them, and removes the 4-way partial-state merge.
// max without the operand canonicalisation fmaxf() implies (the inputs are MFMA results / finite floats or -inf)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

typedef __attribute__((address_space(3))) void lds_void_a;
typedef __attribute__((address_space(1))) const void gbl_void_a;

// KS = 2 ("key split"): 8 waves per block - wave w works on query group w & 3 like before, but only on half w >> 2 (32 keys) of
// every staged 64-key tile, and the two partial softmax states of a query group are merged through the LDS at the end.  At the
// denoiser's shape (n = 870, 32 (batch, head) pairs) the chip holds 1741 sixteen-query chains for 1024 SIMDs: the launch under-fills
// it, and splitting every key tile over two wave groups doubles the waves per SIMD at no extra K / V traffic (the same staged tile
// serves both halves): -2 % on the sampler iteration (profiles/r03_ab_flash_split.txt).  The kernel itself is bound by VALU issue -
// 15.7 VALU instructions per MFMA, VALU pipe ~76 % busy at four waves per SIMD (profiles/r03_pmc_flash_kbench.txt) - so what
// moves it further is fewer softmax instructions per score, not more parallelism.
template <typename T, int NQ, int KS>
__global__ __launch_bounds__(256 * KS, 2) void flash_lds_kernel(FlashArgs a) {
  typedef typename Vec<T>::x8 x8;
  typedef typename Vec<T>::x4 x4;
  constexpr int ST = 3, KT = 64;                 // ring stages, keys per tile
  constexpr int STAGE = 2 * KT * 64;             // elements per stage: K tile [64 keys][64] + V^T tile [64 dims][64 keys]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ring = (T*)smem_raw;                        // [ST][STAGE]
  float* rp = (float*)(ring + ST * STAGE);       // [132] relative-position table (ONE shared object: a second one de-pipelines the ring)
  int bx = blockIdx.x, bh = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int lin = bh * gx + bx, xcd = lin & 7, slot = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int lin2 = xcd * per + min(xcd, rem) + slot;
    bh = lin2 / gx;
    bx = lin2 - bh * gx;
  }
  const int h = bh % a.heads, b = bh / a.heads;
  const int lane = threadIdx.x & 63, wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave_id & 3, kp_own = wave_id >> 2;  // query group of this wave; with KS == 2 the half of every key tile it owns
  const int fr = lane & 15, fg = lane >> 4;
  const int ns = a.n;                                             // row stride of the operands
  const int n = a.nv_period > 0 ? a.nv[b % a.nv_period] : a.n;    // valid keys / queries of this batch row (padded batches)
  const int qblock = bx * 64 * NQ;               // first query of the block
  if (qblock >= n) return;                       // a block of padding queries only (block-uniform, before the first barrier)
  if (a.relpos) {
    if (threadIdx.x < 129) rp[threadIdx.x] = a.relpos[h * 129 + threadIdx.x];
    __syncthreads();
  }
  const int qbase = qblock + wave * 16 * NQ;     // first query of this wave (may be >= n: the wave then only helps loading)
  const T* Q = (const T*)a.q + (size_t)bh * ns * 64;
  const T* K = (const T*)a.k + (size_t)bh * ns * 64;
  const T* VT = (const T*)a.vt + (size_t)bh * 64 * a.n_pad;

  x8 qf[NQ][2];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    const int qr = min(qbase + iq * 16 + fr, n - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[iq][ks] = *(const x8*)(Q + (size_t)qr * 64 + ks * 32 + fg * 8);
  }
  float m_run[NQ], l_run[NQ];
  f32x4 acc[NQ][4];
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    m_run[iq] = -1e30f;
    l_run[iq] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) acc[iq][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int q_last_blk = min(qblock + 64 * NQ, n) - 1;
  const int q_last = min(qbase + 16 * NQ, n) - 1;            // < qbase when the wave has no query
  const int kend_blk = a.causal ? q_last_blk + 1 : n;        // block-uniform: every wave walks the same tiles
  const int kend = qbase >= n ? 0 : (a.causal ? q_last + 1 : n);  // keys this wave's queries can see (none: the wave only helps loading)
  const int ntile = (kend_blk + KT - 1) / KT;

  // stage fill: 16 one-KiB pieces per tile (8 rows x 128 B each): pieces 0-7 = K rows, 8-15 = V^T rows; 4 per wave
  const int lr = lane >> 3, lc = lane & 7;
  auto issue = [&](int t, int stage) {
    const int key0 = t * KT;
    T* base = ring + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4 / KS; ++i) {
      const int piece = wave_id + 4 * KS * i;
      const int row = (piece & 7) * 8 + lr;
      const int chunk = lc ^ ((row >> 1) & 7);
      const T* src;
      if (piece < 8) src = K + (size_t)min(key0 + row, n - 1) * 64 + chunk * 8;                      // key row, 8 dims
      else src = VT + (size_t)row * a.n_pad + min(key0 + chunk * 8, a.n_pad - 8);                    // dim row, 8 keys
      __builtin_amdgcn_global_load_lds((gbl_void_a*)src, (lds_void_a*)(base + piece * 512), 16, 0, 0);
    }
  };

  constexpr float LOG2E = 1.4426950408889634f;
  // one 32-key half tile (kp = 0 / 1) of the staged tile against this wave's queries
  auto process = [&](const T* kt, const T* vt, int kp, int key0) {
    x8 kf[2][2], vf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int r = kp * 32 + kb * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[kb][ks] = *(const x8*)(kt + r * 64 + (((ks * 4 + fg) ^ ((r >> 1) & 7)) * 8));
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int d = blk * 16 + fr;
      const int sw = (d >> 1) & 7;
      // keys kp*32 + fg*4 .. +3 (lo) and +16 .. +19 (hi): chunk = key / 8, 8-byte half (fg & 1)
      const int c_lo = kp * 4 + (fg >> 1), c_hi = c_lo + 2;
      const x4 lo = *(const x4*)(vt + d * 64 + ((c_lo ^ sw) * 8) + (fg & 1) * 4);
      const x4 hi = *(const x4*)(vt + d * 64 + ((c_hi ^ sw) * 8) + (fg & 1) * 4);
      x8 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      vf[blk] = v;
    }
    const bool tail = key0 + 32 > n;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      const int q0 = qbase + iq * 16;
      const int qi = q0 + fr;
      float sv[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
        st = mfma16(kf[kb][0], qf[iq][0], st);
        st = mfma16(kf[kb][1], qf[iq][1], st);
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[kb][r] = st[r];
      }
      // Relative-position bias.  A tile that lies entirely >= 64 positions after (or before) every query of the group gets ONE
      // saturated bucket value for all its scores: a uniform shift, which is folded into the row maximum and the exponent below
      // instead of being added to the 8 scores (most tiles at n = 870).  Only the tiles inside the +-64 window look the table up.
      float cbias = 0.f;
      if (a.relpos) {
        if (key0 - (q0 + 15) >= 64) {          // every key is >= 64 after every query: bucket saturated
          cbias = rp[128];
        } else if (q0 - (key0 + 31) >= 64) {   // every key is >= 64 before every query
          cbias = rp[0];
        } else {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int d = key0 + kb * 16 + fg * 4 + r - qi;
              d = d < -64 ? -64 : (d > 64 ? 64 : d);
              sv[kb][r] += rp[d + 64];
            }
        }
      }
      if (tail || (a.causal && key0 + 31 > q0)) {  // masks only on the last key tile / the causal diagonal
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kb * 16 + fg * 4 + r;
            if (key >= n || (a.causal && key > qi)) sv[kb][r] = -INFINITY;
          }
      }
      // row maximum: v_max3 / v_max straight on the MFMA results (fmaxf() would first canonicalise every operand: 8 extra VALU)
      float mx = vmax3(vmax3(vmax3(sv[0][0], sv[0][1], sv[0][2]), sv[0][3], sv[1][0]), sv[1][1], sv[1][2]);
      mx = vmax(mx, sv[1][3]);
      {
        float x0, x1;
        pair_xor16(mx, x0, x1);
        mx = vmax(x0, x1);
        pair_xor32(mx, x0, x1);
        mx = vmax(x0, x1) + cbias;
      }
      const float m_new = vmax(m_run[iq], mx);
      if (__any(m_new > m_run[iq])) {  // some row maximum moved: rescale the running state (exact when skipped)
        const float alpha = __builtin_amdgcn_exp2f((m_run[iq] - m_new) * LOG2E);
        l_run[iq] *= alpha;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          acc[iq][blk][0] *= alpha; acc[iq][blk][1] *= alpha; acc[iq][blk][2] *= alpha; acc[iq][blk][3] *= alpha;
        }
        m_run[iq] = m_new;
      }
      const float mc = (m_new - cbias) * LOG2E;  // exp2(s * log2e - mc) == exp(s + cbias - m_new)
      float pv[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[kb][r] = __builtin_amdgcn_exp2f(fmaf(sv[kb][r], LOG2E, -mc));
      x8 pf;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[kb * 4 + r] = (T)pv[kb][r];
      const float psum = ((pv[0][0] + pv[0][1]) + (pv[0][2] + pv[0][3])) + ((pv[1][0] + pv[1][1]) + (pv[1][2] + pv[1][3]));
      l_run[iq] += psum;  // lane-partial: reduced over the four key groups once, after the loop
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[iq][blk] = mfma16(vf[blk], pf, acc[iq][blk]);
    }
  };

  constexpr int G = 4 / KS;  // LDS-DMA instructions per wave per tile
  const int last = ntile - 1;
#pragma unroll
  for (int s_ = 0; s_ < ST - 1; ++s_) issue(min(s_, last), s_);
  int slot = 0;
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + ST - 1;
    if (nslot >= ST) nslot -= ST;
    issue(min(t + ST - 1, last), nslot);
    const int key0 = t * KT;
    const T* kt = ring + slot * STAGE;
    const T* vt = kt + KT * 64;
    if constexpr (KS == 1) {
      if (key0 < kend) process(kt, vt, 0, key0);            // wave-uniform: tiles beyond this wave's causal horizon are skipped
      if (key0 + 32 < kend) process(kt, vt, 1, key0 + 32);
    } else {
      if (key0 + kp_own * 32 < kend) process(kt, vt, kp_own, key0 + kp_own * 32);
    }
    slot = slot + 1 == ST ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (KS == 2) {
    // merge the two halves' (m, l, acc) per query row: the upper waves park theirs in the ring (free now), the lower ones combine
    float* mg = (float*)ring;  // [4 query groups][NQ][18][64 lanes]
    __syncthreads();            // every wave is done reading the ring
    if (kp_own == 1) {
#pragma unroll
      for (int iq = 0; iq < NQ; ++iq) {
        float* d = mg + ((size_t)(wave * NQ + iq) * 18) * 64 + lane;
        d[0] = m_run[iq];
        d[64] = l_run[iq];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[(2 + blk * 4 + r) * 64] = acc[iq][blk][r];
      }
    }
    __syncthreads();
    if (kp_own == 1) return;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
      const float* d = mg + ((size_t)(wave * NQ + iq) * 18) * 64 + lane;
      const float m1 = d[0], l1 = d[64];
      const float mm = fmaxf(m_run[iq], m1);
      const float a0 = __builtin_amdgcn_exp2f((m_run[iq] - mm) * LOG2E), a1 = __builtin_amdgcn_exp2f((m1 - mm) * LOG2E);
      l_run[iq] = l_run[iq] * a0 + l1 * a1;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[iq][blk][r] = acc[iq][blk][r] * a0 + d[(2 + blk * 4 + r) * 64] * a1;
    }
  }
#pragma unroll
  for (int iq = 0; iq < NQ; ++iq) {
    l_run[iq] = add_xor16(l_run[iq]);
    l_run[iq] = add_xor32(l_run[iq]);
    const int qi = qbase + iq * 16 + fr;
    if (qi < n) {
      const float inv = 1.0f / l_run[iq];
      T* o = (T*)a.out + ((size_t)b * ns + qi) * a.ldo + h * 64 + fg * 4;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
        *(x4*)(o + blk * 16) = pack4<T>(acc[iq][blk][0] * inv, acc[iq][blk][1] * inv, acc[iq][blk][2] * inv, acc[iq][blk][3] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------- 32-query waves (round 5)
// The same algorithm on v_mfma_f32_32x32x16: a wave owns 32 queries, a workgroup 128 (4 query groups x KS key halves), K / V^T tiles
// of 64 keys staged once per workgroup through the same 3-stage LDS-DMA ring.  Why: flash_lds_kernel issues 15.7 VALU instructions per
// 16x16x32 MFMA (profiles/r03_pmc_flash_kbench.txt) - per 16 queries x 32 keys a lane holds 8 scores, and everything that is per ROW
// (cross-lane maximum, running-state update, the relative-position window test, the LDS fragment addresses) is paid per 8 scores.  Here
// a lane holds ONE query (l & 31) and 16 scores of it per 32-key block; its row maximum is 15 in-lane v_max + one permlane32 swap, the
// fragment reads serve twice the flops (a K fragment feeds a 32 x 32 x 16 product), and the P^T operand of the second product is again
// the score registers in the order the first product left them: lane half h = l >> 5 holds keys 4 h + 8 i + j of the block (reg 4 i + j),
// so k-slot e of PV step kk is score register 8 kk + e, and V^T is read with that permutation (two 8-byte reads per fragment).
template <typename T, int KS>
__global__ __launch_bounds__(256 * KS, KS == 4 ? 4 : 2 * KS) void flash32_kernel(FlashArgs a) {
  typedef typename Vec<T>::x8 x8;
  typedef typename Vec<T>::x4 x4;
  // (KS = 4: 16 waves, tiles of 128 keys, every wave one 32-key block of each - built and measured equal to KS = 2 at the denoiser's shape,
  //  profiles/r05_ab_flash_ks4_and_epilogue_prefetch.txt; not instantiated)
  constexpr int ST = 3, KT = KS == 4 ? 128 : 64;
  constexpr int STAGE = 2 * KT * 64;
  constexpr int CPR = KT / 8, RPP = 64 / CPR;   // V^T tile: 16-byte chunks per row (64 d-rows of KT keys), d-rows per 1-KiB piece
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ring = (T*)smem_raw;
  float* rp = (float*)(ring + ST * STAGE);
  int bx = blockIdx.x, bh = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int lin = bh * gx + bx, xcd = lin & 7, slot = lin >> 3;
    const int per = total >> 3, rem = total & 7;
    const int lin2 = xcd * per + min(xcd, rem) + slot;
    bh = lin2 / gx;
    bx = lin2 - bh * gx;
  }
  const int h = bh % a.heads, b = bh / a.heads;
  const int lane = threadIdx.x & 63, wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave_id & 3, kp_own = wave_id >> 2;
  const int ql = lane & 31, hh = lane >> 5;
  const int ns = a.n;
  const int n = a.nv_period > 0 ? a.nv[b % a.nv_period] : a.n;
  const int qblock = bx * 128;
  if (qblock >= n) return;
  // relative-position table with the saturated buckets extended by 64 entries on both sides: rp[i] = bias(clamp(i - 128, -64, 64)), so a
  // tile that straddles the +-64 window reads rp[key - query + 128] without clamping (|key - query| <= 125 there)
  if (a.relpos) {
    for (int i = threadIdx.x; i < 257; i += 256 * KS) rp[i] = a.relpos[h * 129 + min(max(i - 64, 0), 128)];
    __syncthreads();
  }
  const int qbase = qblock + wave * 32;
  const T* Q = (const T*)a.q + (size_t)bh * ns * 64;
  const T* K = (const T*)a.k + (size_t)bh * ns * 64;
  const T* VT = (const T*)a.vt + (size_t)bh * 64 * a.n_pad;

  x8 qf[4];
  {
    const int qr = min(qbase + ql, n - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const x8*)(Q + (size_t)qr * 64 + ks * 16 + hh * 8);
  }
  float m_run = -1e30f, l_run = 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[db][v] = 0.f;
  const int q_last_blk = min(qblock + 128, n) - 1;
  const int q_last = min(qbase + 32, n) - 1;
  const int kend_blk = a.causal ? q_last_blk + 1 : n;
  const int kend = qbase >= n ? 0 : (a.causal ? q_last + 1 : n);
  const int ntile = (kend_blk + KT - 1) / KT;

  const int lr = lane >> 3, lc = lane & 7;
  constexpr int G = (KT / 4) / (4 * KS);  // LDS-DMA pieces per wave per tile: KT / 8 pieces of K + KT / 8 of V^T over 4 KS waves
  auto issue = [&](int t, int stage) {
    const int key0 = t * KT;
    T* base = ring + stage * STAGE;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int piece = wave_id + 4 * KS * i;
      const T* src;
      if (piece < KT / 8) {                      // K rows: 8 rows x 8 chunks per piece
        const int row = piece * 8 + lr;
        src = K + (size_t)min(key0 + row, n - 1) * 64 + (lc ^ ((row >> 1) & 7)) * 8;
      } else {                                   // V^T rows: RPP rows x CPR chunks per piece, the low 3 chunk bits swizzled
        const int row = (piece - KT / 8) * RPP + lane / CPR, pc = lane % CPR;
        const int chunk = (pc & ~7) | ((pc & 7) ^ ((row >> 1) & 7));
        src = VT + (size_t)row * a.n_pad + min(key0 + chunk * 8, a.n_pad - 8);
      }
      __builtin_amdgcn_global_load_lds((gbl_void_a*)src, (lds_void_a*)(base + piece * 512), 16, 0, 0);
    }
  };

  constexpr float LOG2E = 1.4426950408889634f;
  const int qi = qbase + ql;
  // one 32-key block (kp = 0 / 1 of the staged tile; key0 = its first key) against this wave's 32 queries
  auto process = [&](const T* kt, const T* vt, int kp, int key0) {
    x8 kf[4], vf[2][2];
    {
      const int r = kp * 32 + ql;
      const int sw = (r >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const x8*)(kt + r * 64 + (((ks * 2 + hh) ^ sw) * 8));
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int d = db * 32 + ql;
      const int sw = (d >> 1) & 7;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int c = kp * 4 + 2 * kk;  // 8-key chunk of keys 16 kk .. +7 of the block; this lane half takes keys 4 hh .. +3 of it and of the next
        const x4 lo = *(const x4*)(vt + d * KT + (((c & ~7) | ((c & 7) ^ sw)) * 8) + hh * 4);
        const x4 hi = *(const x4*)(vt + d * KT + ((((c + 1) & ~7) | (((c + 1) & 7) ^ sw)) * 8) + hh * 4);
        x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        vf[db][kk] = v;
      }
    }
    // score register v <-> key key0 + 8 (v >> 2) + 4 hh + (v & 3).  Relative-position bias: a block entirely >= 64 positions after (or
    // before) every query of the group gets ONE saturated bucket value for all its scores - folded into the row maximum and the exponent
    // below; a block that straddles the window takes its 16 bias values as the ACCUMULATOR INPUT of the first product (S^T = K Q^T + C), so
    // no score register is touched after the MFMA chain on either path (a post-MFMA add made hipcc copy all 16 registers out and back on
    // the common, saturated path).
    float cbias = 0.f;
    f32x16 st;
    const bool window = a.relpos != nullptr && key0 - (qbase + 31) < 64 && qbase - (key0 + 31) < 64;  // wave-uniform
    if (window) {
      const float* rb = rp + (key0 - qi + 4 * hh + 128);  // one address per lane, 16 reads at constant offsets
#pragma unroll
      for (int v = 0; v < 16; ++v) st[v] = rb[8 * (v >> 2) + (v & 3)];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = mfma32(kf[ks], qf[ks], st);
    } else {
      if (a.relpos) cbias = key0 - (qbase + 31) >= 64 ? rp[256] : rp[0];
#pragma unroll
      for (int v = 0; v < 16; ++v) st[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = mfma32(kf[ks], qf[ks], st);
    }
    float sv[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) sv[v] = st[v];
    if (key0 + 32 > n || (a.causal && key0 + 31 > qbase)) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int key = key0 + 8 * (v >> 2) + 4 * hh + (v & 3);
        if (key >= n || (a.causal && key > qi)) sv[v] = -INFINITY;
      }
    }
    float mx = vmax3(sv[0], sv[1], sv[2]);
    mx = vmax3(mx, sv[3], sv[4]);
    mx = vmax3(mx, sv[5], sv[6]);
    mx = vmax3(mx, sv[7], sv[8]);
    mx = vmax3(mx, sv[9], sv[10]);
    mx = vmax3(mx, sv[11], sv[12]);
    mx = vmax3(mx, sv[13], sv[14]);
    mx = vmax(mx, sv[15]);
    {
      float x0, x1;
      pair_xor32(mx, x0, x1);
      mx = vmax(x0, x1) + cbias;
    }
    const float m_new = vmax(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[db][v] *= alpha;
      m_run = m_new;
    }
    const float mc = (m_new - cbias) * LOG2E;
    float pv[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) pv[v] = __builtin_amdgcn_exp2f(fmaf(sv[v], LOG2E, -mc));
    x8 pf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[kk][e] = (T)pv[kk * 8 + e];
    l_run += (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
             (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int db = 0; db < 2; ++db) acc[db] = mfma32(vf[db][kk], pf[kk], acc[db]);
  };

  const int last = ntile - 1;
#pragma unroll
  for (int s_ = 0; s_ < ST - 1; ++s_) issue(min(s_, last), s_);
  int slot = 0;
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + ST - 1;
    if (nslot >= ST) nslot -= ST;
    issue(min(t + ST - 1, last), nslot);
    const int key0 = t * KT;
    const T* kt = ring + slot * STAGE;
    const T* vt = kt + KT * 64;
    if constexpr (KS == 1) {
      if (key0 < kend) process(kt, vt, 0, key0);
      if (key0 + 32 < kend) process(kt, vt, 1, key0 + 32);
    } else {
      if (key0 + kp_own * 32 < kend) process(kt, vt, kp_own, key0 + kp_own * 32);
    }
    slot = slot + 1 == ST ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (KS >= 2) {
    // merge the key splits' (m, l, acc) per query row through the ring (free now): the upper half of the splits parks, the lower half combines;
    // KS = 4 does that twice (splits 2, 3 into 0, 1, then 1 into 0)
    float* mg = (float*)ring;  // [KS / 2][4 query groups][34][64 lanes]
#pragma unroll
    for (int half = KS / 2; half >= 1; half >>= 1) {
      __syncthreads();  // every wave is done with the ring / with the previous round's scratch
      if (kp_own >= half && kp_own < 2 * half) {
        float* d = mg + ((size_t)((kp_own - half) * 4 + wave) * 34) * 64 + lane;
        d[0] = m_run;
        d[64] = l_run;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int v = 0; v < 16; ++v) d[(2 + db * 16 + v) * 64] = acc[db][v];
      }
      __syncthreads();
      if (kp_own < half) {
        const float* d = mg + ((size_t)(kp_own * 4 + wave) * 34) * 64 + lane;
        const float m1 = d[0], l1 = d[64];
        const float mm = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run - mm) * LOG2E), a1 = __builtin_amdgcn_exp2f((m1 - mm) * LOG2E);
        l_run = l_run * a0 + l1 * a1;
        m_run = mm;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[db][v] = acc[db][v] * a0 + d[(2 + db * 16 + v) * 64] * a1;
      }
    }
    if (kp_own != 0) return;
  }
  l_run = add_xor32(l_run);
  if (qi < n) {
    const float inv = 1.0f / l_run;
    T* o = (T*)a.out + ((size_t)b * ns + qi) * a.ldo + h * 64 + hh * 4;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(x4*)(o + db * 32 + i * 8) = pack4<T>(acc[db][4 * i] * inv, acc[db][4 * i + 1] * inv, acc[db][4 * i + 2] * inv, acc[db][4 * i + 3] * inv);
  }
}

template <typename T, int KS>
static int launch_flash32(const ProfScope& ps, const FlashArgs& a, hipStream_t stream) {
  constexpr int KT = KS == 4 ? 128 : 64;
  constexpr int smem = 3 * 2 * KT * 64 * 2 + 260 * 4;
  static_assert(KS == 1 || (KS / 2) * 4 * 34 * 64 * 4 <= 3 * 2 * KT * 64 * 2, "the merge scratch must fit the ring");
  static bool attr_set = false;
  if (!attr_set) {
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)flash32_kernel<T, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  dim3 grid(cdiv(a.n, 128), a.BH);
  launch_timed(ps, flash32_kernel<T, KS>, grid, dim3(256 * KS), smem, stream, a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int NQ, int KS>
static int launch_flash_lds(const ProfScope& ps, const FlashArgs& a, hipStream_t stream) {
  constexpr int smem = 3 * 2 * 64 * 64 * 2 + 132 * 4;
  static_assert(KS == 1 || 4 * NQ * 18 * 64 * 4 <= 3 * 2 * 64 * 64 * 2, "the merge scratch must fit the ring");
  static bool attr_set = false;
  if (!attr_set) {
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)flash_lds_kernel<T, NQ, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  dim3 grid(cdiv(a.n, 64 * NQ), a.BH);
  launch_timed(ps, flash_lds_kernel<T, NQ, KS>, grid, dim3(256 * KS), smem, stream, a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

bool g_flash32 = true;  // tt_flash_variant: 0 = the 16-query-wave kernels everywhere (A/B runs)

int flash_attention_launch(int dtype, const FlashArgs& a, hipStream_t stream) {
  if (dtype == DT_F32) return flash_f32_launch(a, stream);  // verification mode (attention_f32.hip)
  TT_REQUIRE(a.BH > 0 && a.n > 0 && a.heads > 0 && a.BH % a.heads == 0, "flash: bad shape BH=%d n=%d heads=%d", a.BH, a.n, a.heads);
  TT_REQUIRE(a.n_pad % 32 == 0 && a.n_pad >= ((a.n + 31) / 32) * 32, "flash: n_pad=%d must be a multiple of 32 covering n=%d", a.n_pad, a.n);
  TT_REQUIRE(a.ldo % 4 == 0, "flash: ldo must be a multiple of 4");
  // 32 queries per wave once there is enough work to fill the chip; 16 otherwise.
  // QK^T + PV: 4 * n * n * 64 flops per (batch, head) (halved when causal); Q, K, V read + O written once
  ProfScope ps(PROF_FLASH, stream, 4.0 * a.BH * (double)a.n * a.n * 64 * (a.causal ? 0.5 : 1.0), 4.0 * a.BH * (double)a.n * 64 * 2.0, true);
  if (a.n > 128 && !a.causal && a.variant != 2 && g_flash32) {
    // 32-query waves on v_mfma_f32_32x32x16 (flash32_kernel), 128 queries per workgroup; launches of fewer than ~2 workgroups per CU
    // split every key tile over two wave groups (the denoiser: 7 x 32 workgroups)
    const long blocks128 = (long)cdiv(a.n, 128) * a.BH;
    if (blocks128 < 512 && a.variant != 1) return dtype == DT_BF16 ? launch_flash32<bf16, 2>(ps, a, stream) : launch_flash32<f16, 2>(ps, a, stream);
    return dtype == DT_BF16 ? launch_flash32<bf16, 1>(ps, a, stream) : launch_flash32<f16, 1>(ps, a, stream);
  }
  if (a.n > 128) {
    // LDS-staged kernel: 64 queries per block while that keeps >= 2 blocks per CU busy, 128 otherwise (half the K / V traffic per
    // flop; measured on the kbench shapes: 32 queries per wave only pays from ~2048 blocks of 64 queries on)
    const long blocks64 = (long)cdiv(a.n, 64) * a.BH;
    if (blocks64 >= 2048) return dtype == DT_BF16 ? launch_flash_lds<bf16, 2, 1>(ps, a, stream) : launch_flash_lds<f16, 2, 1>(ps, a, stream);
    // fewer than ~4 blocks per CU: the launch is paid for the per-wave dependency chain - split every key tile over two wave groups
    // (in-situ A/B, profiles/r03_ab_flash_split.txt: denoiser iteration 1.555 -> 1.520 ms; the softmax VALU diet of this round -4 % per launch)
    if (blocks64 < 1024 && a.variant != 1) return dtype == DT_BF16 ? launch_flash_lds<bf16, 1, 2>(ps, a, stream) : launch_flash_lds<f16, 1, 2>(ps, a, stream);
    return dtype == DT_BF16 ? launch_flash_lds<bf16, 1, 1>(ps, a, stream) : launch_flash_lds<f16, 1, 1>(ps, a, stream);
  }
  // short sequences (prefill of a few dozen rows, reduced test configurations): the register-prefetch kernel, the 4 waves of a
  // block share one 16-query block and split the keys
  dim3 grid(cdiv(a.n, 16), a.BH);
  if (dtype == DT_BF16) launch_timed(ps, flash_kernel<bf16, 1, true>, grid, dim3(256), 0, stream, a);
  else launch_timed(ps, flash_kernel<f16, 1, true>, grid, dim3(256), 0, stream, a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------- decode
// 8-wide dot product with fp32 accumulation on the packed-pair dot instructions (v_dot2c_f32_bf16 / v_dot2c_f32_f16):
// the query stays packed (32 VGPRs instead of 64 floats) and a key costs 32 VALU ops instead of 64 converts + 64 FMAs.
__device__ __forceinline__ float dot8(Vec<bf16>::x8 a, Vec<bf16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}
__device__ __forceinline__ float dot8(Vec<f16>::x8 a, Vec<f16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}

// One wave per (sequence, head); 4 waves per block.  Sized to <= 128 VGPRs so that all B*heads = 4096 waves of the
// full candidate batch are resident at once (16 waves per CU): with 3 blocks per CU the 1024 blocks ran as a full
// round plus a quarter-full tail.  The shared prefix and the sequence's own keys are walked as separate, uniform
// segments: every load is then (wave-uniform base) + (32-bit lane offset) - no 64-bit address pairs held in VGPRs -
// and unconditional (clamped key index), because a branch between two groups of loads makes the compiler drain the
// first group before it issues the second.
constexpr int DEC_VROWS = 4;             // V key rows per lane per register set (two sets in flight)
constexpr int DEC_VKEYS = 8 * DEC_VROWS;  // keys per wave per PV iteration: 8 key sub-rows x DEC_VROWS

template <typename T>
__global__ __launch_bounds__(256, 4) void decode_attn_kernel(DecodeAttnArgs a, int ctx_cap) {  // 4 waves per SIMD => <= 128 VGPRs
  typedef typename Vec<T>::x8 x8;
  extern __shared__ __attribute__((aligned(16))) float sc_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = min((int)blockIdx.x * 4 + wave, a.B * a.heads - 1);  // surplus waves of the last block repeat its last pair
  const int b = pair / a.heads;
  const int h = pair % a.heads;
  const int tgen = *a.step + 1;       // generated keys 0..*step
  const int P1 = a.P1;
  const int ctx = P1 + tgen;
  float* sc = sc_all + (size_t)wave * ctx_cap;   // scores: [0, P1) prefix keys, [P1, ctx) own keys

  const T* kp = (const T*)a.kp + (size_t)h * P1 * 64;
  const T* vp = (const T*)a.vp + (size_t)h * P1 * 64;
  const size_t bh = (size_t)b * a.heads + h;
  const T* kc = (const T*)a.kc + bh * 8 * a.tmax * 8;
  const T* vc = (const T*)a.vc + bh * a.tmax * 64;

  float mx = -1e30f;
  {
    x8 qk[8];
    const T* qp = (const T*)a.q + (size_t)b * a.heads * 64 + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) qk[c] = *(const x8*)(qp + c * 8);
    // Lane-per-key dot products, TWO key slots (16 x 16-byte loads) in flight per lane per iteration.  Slot list:
    // prefix keys in 64-key slots (row-major rows of 64), then own keys in 64-key slots (chunk-major [8][tmax][8]).
    const int nsp = (P1 + 63) >> 6, nso = (tgen + 63) >> 6;
#pragma unroll 1
    for (int sl0 = 0; sl0 < nsp + nso; sl0 += 2) {
      x8 kk[2][8];
      int key[2];
      bool live[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int slot = min(sl0 + u, nsp + nso - 1);   // an odd slot count repeats the last slot (cached), result dropped
        const bool pre = slot < nsp;                    // wave-uniform
        const int k = (pre ? slot : slot - nsp) * 64 + lane;
        const int lim = pre ? P1 : tgen;
        const int kcl = min(k, lim - 1);
        const char* base = (const char*)(pre ? kp : kc);
        const unsigned off = (pre ? (unsigned)kcl * 64u : (unsigned)kcl * 8u) * (unsigned)sizeof(T);  // byte offsets, 32-bit
        const unsigned cs = (pre ? 8u : (unsigned)a.tmax * 8u) * (unsigned)sizeof(T);
#pragma unroll
        for (int c = 0; c < 8; ++c) kk[u][c] = *(const x8*)(base + (off + c * cs));
        key[u] = (pre ? 0 : P1) + k;
        live[u] = k < lim && sl0 + u < nsp + nso;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) sv = dot8(qk[c], kk[u][c], sv);
        if (live[u]) {
          sc[key[u]] = sv;
          mx = fmaxf(mx, sv);
        }
      }
    }
  }

  // PV: a V row is 128 bytes, read as 8 lanes x 16 bytes, so a wave instruction covers 8 whole key rows (1 KiB, like the K
  // loads; 8-byte loads moved half as much per instruction and measured 6 % slower).  Lane -> (key sub-row kk8 = lane >> 3,
  // channel group cg = lane & 7: 8 channels); iteration `it` covers DEC_VKEYS keys of one segment.
  const int kk8 = lane >> 3, cg = lane & 7;
  const int nvp = (P1 + DEC_VKEYS - 1) / DEC_VKEYS, nvo = (tgen + DEC_VKEYS - 1) / DEC_VKEYS;
  const int nit = nvp + nvo;
  auto load_v = [&](x8 (&t)[DEC_VROWS], int it) {
    const int itc = min(it, nit - 1);  // past the end: repeat the last iteration's rows (cached), weighted 0
    const bool pre = itc < nvp;
    const char* base = (const char*)(pre ? vp : vc);
    const int k0 = (pre ? itc : itc - nvp) * DEC_VKEYS + kk8, lim = pre ? P1 : tgen;
#pragma unroll
    for (int u = 0; u < DEC_VROWS; ++u) {
      const unsigned jc = (unsigned)min(k0 + 8 * u, lim - 1);
      t[u] = *(const x8*)(base + (jc * 64u + (unsigned)cg * 8u) * (unsigned)sizeof(T));  // uniform base + 32-bit byte offset
    }
  };
  // the first V rows do not depend on the scores: request them before the softmax
  x8 ta[DEC_VROWS], tb[DEC_VROWS];
  load_v(ta, 0);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < ctx; j += 64) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();  // every lane's sc[] writes are visible to the whole wave (and block)
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = 0.f;
  auto consume = [&](const x8 (&t)[DEC_VROWS], int it) {
    const bool pre = it < nvp;
    const int k0 = (pre ? it : it - nvp) * DEC_VKEYS + kk8, lim = it < nit ? (pre ? P1 : tgen) : 0;
    const float* scs = sc + (pre ? 0 : P1);
#pragma unroll
    for (int u = 0; u < DEC_VROWS; ++u) {
      const int j = k0 + 8 * u;
      const float pj = j < lim ? scs[j] : 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] += pj * (float)t[u][c];
    }
  };
#pragma unroll 1
  for (int it = 0; it < nit; it += 2) {  // two register sets: the next rows are in flight while these are summed
    load_v(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);  // fences keep exactly two row sets live (an early third set spills)
    consume(ta, it);
    __builtin_amdgcn_sched_barrier(0);
    load_v(ta, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    consume(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {  // sum the 8 key sub-rows (lanes with equal channel group)
    o[c] = add_xor8(o[c]);
    o[c] = add_xor16(o[c]);
    o[c] = add_xor32(o[c]);
  }
  if ((int)blockIdx.x * 4 + wave < a.B * a.heads && kk8 == 0) {
    const float inv = 1.0f / sum;
    x8 r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = (T)(o[c] * inv);
    *(x8*)((T*)a.out + (size_t)b * a.heads * 64 + h * 64 + cg * 8) = r;
  }
}

// Shared-prefix variant.  Every candidate of an utterance attends to the SAME [cond | text | start] prefix keys; the kernel
// above reads them once per (sequence, head) wave - 15 KB per wave, 61 MB of L2 -> CU traffic per launch at 256 candidates,
// which is on the critical path of every wave (per-CU L2 bandwidth is ~50 GB/s) although it never touches HBM.  Here a
// workgroup is NSEQ waves = NSEQ sequences of ONE head: the head's prefix K / V are staged into LDS once per workgroup
// (global_load_lds; K re-laid chunk-major on the fly through the per-lane source address so the lane-per-key reads are
// conflict-free ds_read_b128), and only the per-sequence cache is streamed from HBM.  The first own-key slots are requested
// before the workgroup waits for the staged prefix (counted vmcnt: the direct-to-LDS loads are older in the queue), so the
// HBM stream starts at once.  Arithmetic and summation order per (sequence, head) are exactly those of decode_attn_kernel.
// -DTT_ATTN_STAMPS (a variant build, scripts/attn_phases.py): wave 0 of every workgroup keeps the 100 MHz wall clock of its phase
// boundaries in scalar registers (no vector-memory operation: the counted vmcnt waits are untouched) and files them at the end
#ifdef TT_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[4096][10];
#define TT_ASTAMP(i) do { st[i] = wall_clock64(); } while (0)
#else
#define TT_ASTAMP(i)
#endif
template <typename T, int NSEQ>
__global__ __launch_bounds__(NSEQ * 64, 4) void decode_attn_lds_kernel(DecodeAttnArgs a, int ctx_cap, int kl_bytes, int vl_bytes) {
  typedef typename Vec<T>::x8 x8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dec[];
#ifdef TT_ATTN_STAMPS
  unsigned long long st[10];
#endif
  TT_ASTAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tgen = *a.step + 1;       // generated keys 0..*step (read first: a scalar load, nothing in front of it to drain)
  const int h = (int)blockIdx.x;      // grid = (heads, sequence groups)
  const int b_raw = (int)blockIdx.y * NSEQ + wave;
  const int b = min(b_raw, a.B - 1);  // surplus waves of the last group repeat its last sequence (never stored)
  int P1 = a.P1;
  size_t goff = 0;
  if (a.ngroups > 1) {  // several utterances in one batch: this workgroup's sequences all belong to one of them (group_size % NSEQ == 0)
    const int grp = ((int)blockIdx.y * NSEQ) / a.group_size;
    P1 = a.p1_tab[grp];
    goff = (size_t)grp * a.prefix_group_stride;
  }
  const T* kp = (const T*)a.kp + goff + (size_t)h * P1 * 64;
  const T* vp = (const T*)a.vp + goff + (size_t)h * P1 * 64;
  unsigned char* Kl = smem_dec;                 // [slot][8 chunks][64 keys][8]  (chunk-major like the per-sequence cache)
  unsigned char* Vl = smem_dec + kl_bytes;      // [key][64]
  float* sc = (float*)(smem_dec + kl_bytes + vl_bytes) + (size_t)wave * ctx_cap;
  const int nsp = (P1 + 63) >> 6;

  // stage the prefix: K instruction (slot s, chunk c): lane k <- kp[s*64 + k][c*8 .. c*8+7]; V instruction i: rows 8i .. 8i+7
  {
    const int nk_ins = nsp * 8, nv_ins = (P1 + 7) >> 3;
    for (int i = wave; i < nk_ins; i += NSEQ) {
      const int sidx = i >> 3, c = i & 7;
      const int key = min(sidx * 64 + lane, P1 - 1);
      __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)((const char*)kp + ((size_t)key * 64 + c * 8) * sizeof(T)),
                                       (__attribute__((address_space(3))) void*)(Kl + (size_t)i * 1024), 16, 0, 0);
    }
    for (int i = wave; i < nv_ins; i += NSEQ) {
      const int row = min(i * 8 + (lane >> 3), P1 - 1);
      __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)((const char*)vp + ((size_t)row * 64 + (lane & 7) * 8) * sizeof(T)),
                                       (__attribute__((address_space(3))) void*)(Vl + (size_t)i * 1024), 16, 0, 0);
    }
  }
  const int ctx = P1 + tgen;
  const size_t bh = (size_t)b * a.heads + h;
  const T* kc = (const T*)a.kc + bh * 8 * a.tmax * 8;
  const T* vc = (const T*)a.vc + bh * a.tmax * 64;

  float mx = -1e30f;
  // The query is wave-uniform: it lives in 32 SGPRs (two s_load_dwordx16), not in 32 VGPRs per lane, and its load is not on
  // the vector-memory counter, so nothing the compiler places between the staged prefix and the first use of q can force a
  // vmcnt(0) that would also drain the own-key requests below.  (Inline asm: hipcc only emits scalar loads for memory it can
  // prove read-only, and q was written by the previous kernel.)
  typedef int int16v __attribute__((ext_vector_type(16)));
  typedef int int4v __attribute__((ext_vector_type(4)));
  int16v qlo, qhi;
  {
    const T* qp = (const T*)a.q + (size_t)b * a.heads * 64 + h * 64;
    // early-clobber outputs: the second load still reads the address pair after the first one has been issued (and may have landed)
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(qlo), "=&s"(qhi) : "s"(qp) : "memory");
  }
  const int nso = (tgen + 63) >> 6;
  auto load_own = [&](x8 (&kk)[8], int slot) {  // own keys of slot (clamped: an odd slot count repeats the last slot, result dropped)
    const int k = min(slot, nso - 1) * 64 + lane;
    const unsigned off = (unsigned)min(k, tgen - 1) * 8u * (unsigned)sizeof(T);
    const unsigned cs = (unsigned)a.tmax * 8u * (unsigned)sizeof(T);
#pragma unroll
    for (int c = 0; c < 8; ++c) kk[c] = *(const x8*)((const char*)kc + (off + c * cs));
  };
  x8 k0[8], k1[8];
  load_own(k0, 0);
  load_own(k1, 1);
  TT_ASTAMP(1);
  // the staged prefix must have landed (this wave's direct-to-LDS loads are older than the 16 register loads above)
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(qlo), "+s"(qhi)::"memory");  // q has landed (every later use depends on this statement)
  TT_ASTAMP(2);
  x8 qk[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int16v& src = c < 4 ? qlo : qhi;
    int4v w;
    w[0] = src[(c & 3) * 4 + 0]; w[1] = src[(c & 3) * 4 + 1]; w[2] = src[(c & 3) * 4 + 2]; w[3] = src[(c & 3) * 4 + 3];
    qk[c] = __builtin_bit_cast(x8, w);
  }
  // prefix scores from LDS
#pragma unroll 1
  for (int sidx = 0; sidx < nsp; ++sidx) {
    float sv = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) sv = dot8(qk[c], *(const x8*)(Kl + ((size_t)(sidx * 8 + c) * 64 + lane) * 16), sv);
    const int k = sidx * 64 + lane;
    if (k < P1) {
      sc[k] = sv;
      mx = fmaxf(mx, sv);
    }
  }
  TT_ASTAMP(3);
  // own scores, two slots per iteration
#pragma unroll 1
  for (int sl0 = 0; sl0 < nso; sl0 += 2) {
    if (sl0 > 0) {
      load_own(k0, sl0);
      load_own(k1, sl0 + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s0 = dot8(qk[c], k0[c], s0);
#pragma unroll
    for (int c = 0; c < 8; ++c) s1 = dot8(qk[c], k1[c], s1);
    const int ka = sl0 * 64 + lane, kb = ka + 64;
    if (ka < tgen) {
      sc[P1 + ka] = s0;
      mx = fmaxf(mx, s0);
    }
    if (kb < tgen && sl0 + 1 < nso) {
      sc[P1 + kb] = s1;
      mx = fmaxf(mx, s1);
    }
  }

  TT_ASTAMP(4);
  const int kk8 = lane >> 3, cg = lane & 7;
  const int nvp = (P1 + DEC_VKEYS - 1) / DEC_VKEYS, nvo = (tgen + DEC_VKEYS - 1) / DEC_VKEYS;
  auto load_v = [&](x8 (&t)[DEC_VROWS], int it) {  // own rows of iteration `it` (past the end: the last rows again, weighted 0)
    const int k0r = min(it, nvo - 1) * DEC_VKEYS + kk8;
#pragma unroll
    for (int u = 0; u < DEC_VROWS; ++u) {
      const unsigned jc = (unsigned)min(k0r + 8 * u, tgen - 1);
      t[u] = *(const x8*)((const char*)vc + (jc * 64u + (unsigned)cg * 8u) * (unsigned)sizeof(T));
    }
  };
  x8 ta[DEC_VROWS], tb[DEC_VROWS];
  load_v(ta, 0);  // the first V rows do not depend on the scores: request them before the softmax
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < ctx; j += 64) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // sc[] is private to this wave: LDS operations of a wave execute in order
  TT_ASTAMP(5);
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = 0.f;
  // prefix rows from LDS
#pragma unroll 1
  for (int it = 0; it < nvp; ++it) {
    const int k0r = it * DEC_VKEYS + kk8;
#pragma unroll
    for (int u = 0; u < DEC_VROWS; ++u) {
      const int j = k0r + 8 * u;
      const float pj = j < P1 ? sc[j] : 0.f;
      const x8 t = *(const x8*)(Vl + ((size_t)min(j, P1 - 1) * 64 + cg * 8) * sizeof(T));
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] += pj * (float)t[c];
    }
  }
  TT_ASTAMP(6);
  auto consume = [&](const x8 (&t)[DEC_VROWS], int it) {
    const int k0r = it * DEC_VKEYS + kk8, lim = it < nvo ? tgen : 0;
    const float* scs = sc + P1;
#pragma unroll
    for (int u = 0; u < DEC_VROWS; ++u) {
      const int j = k0r + 8 * u;
      const float pj = j < lim ? scs[j] : 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] += pj * (float)t[u][c];
    }
  };
#pragma unroll 1
  for (int it = 0; it < nvo; it += 2) {  // two register sets: the next rows are in flight while these are summed
    load_v(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    consume(ta, it);
    __builtin_amdgcn_sched_barrier(0);
    load_v(ta, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    consume(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  TT_ASTAMP(7);
#pragma unroll
  for (int c = 0; c < 8; ++c) {  // sum the 8 key sub-rows (lanes with equal channel group)
    o[c] = add_xor8(o[c]);
    o[c] = add_xor16(o[c]);
    o[c] = add_xor32(o[c]);
  }
  if (b_raw < a.B && kk8 == 0) {
    const float inv = 1.0f / sum;
    x8 r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = (T)(o[c] * inv);
    *(x8*)((T*)a.out + (size_t)b * a.heads * 64 + h * 64 + cg * 8) = r;
  }
#ifdef TT_ATTN_STAMPS
  TT_ASTAMP(8);
  if (threadIdx.x == 0) {
    const int wg = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (wg < 4096) {
      for (int i = 0; i < 9; ++i) g_attn_stamps[wg][i] = st[i];
      unsigned xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_attn_stamps[wg][9] = xcc;
    }
  }
#endif
}
#ifdef TT_ATTN_STAMPS
}  // namespace tt
extern "C" int ttx_attn_stamps(unsigned long long* out, int nwg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tt::g_attn_stamps), (size_t)nwg * 10 * sizeof(unsigned long long));
}
namespace tt {
#endif

int decode_attention_launch(int dtype, const DecodeAttnArgs& a, hipStream_t stream) {
  if (dtype == DT_F32) return decode_attn_f32_launch(a, stream);  // verification mode (attention_f32.hip)
  TT_REQUIRE(a.B > 0 && a.heads > 0 && a.P1 >= 0 && a.tmax > 0, "decode_attention: bad shape");
  const int ctx_cap = a.P1 + a.tmax;
  // algorithmic bytes: every sequence reads its own generated K and V rows once (host_tgen keys) + the shared prefix once
  ProfScope ps(PROF_DECODE_ATTN, stream, 4.0 * a.B * a.heads * 64.0 * (a.P1 + a.host_tgen),
               ((double)a.B * a.host_tgen + a.P1) * a.heads * 64 * 2 * 2.0 + 2.0 * a.B * a.heads * 64 * 2.0, true);
  // shared-prefix kernel with 4 sequences per workgroup (measured 2 % ahead of 16 at 256 candidates and 40 % ahead at 32:
  // more, smaller workgroups); the per-wave kernel only when the staged prefix + score rows do not fit the LDS (very long prompts)
  int nseq = a.variant == 1 ? 0 : a.variant == 2 ? 16 : 4;
  if (a.ngroups > 1) {
    TT_REQUIRE(a.ngroups <= 16 && a.group_size > 0 && a.group_size % 4 == 0 && a.B == a.ngroups * a.group_size,
               "decode_attention: %d groups of %d sequences (a multiple of 4) do not make %d sequences", a.ngroups, a.group_size, a.B);
    nseq = 4;
  }
  const int kl_bytes = ((a.P1 + 63) >> 6) * 8 * 1024, vl_bytes = ((a.P1 + 7) >> 3) * 1024;
  if (a.P1 < 1) nseq = 0;
  if (nseq && (size_t)kl_bytes + vl_bytes + (size_t)nseq * ctx_cap * sizeof(float) > 160 * 1024) nseq = nseq == 16 ? 4 : 0;
  if (nseq && (size_t)kl_bytes + vl_bytes + (size_t)nseq * ctx_cap * sizeof(float) > 160 * 1024) nseq = 0;
  if (nseq) {
    const size_t smem = (size_t)kl_bytes + vl_bytes + (size_t)nseq * ctx_cap * sizeof(float);
    const dim3 blocks(a.heads, cdiv(a.B, nseq));
#define TT_DEC(T, NS)                                                                                                              \
    do {                                                                                                                             \
      static bool attr_done = false;                                                                                                 \
      if (!attr_done) {                                                                                                              \
        TT_CHECK_HIP(hipFuncSetAttribute((const void*)decode_attn_lds_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        attr_done = true;                                                                                                            \
      }                                                                                                                              \
      launch_timed(ps, decode_attn_lds_kernel<T, NS>, blocks, dim3(NS * 64), smem, stream, a, ctx_cap, kl_bytes, vl_bytes);          \
    } while (0)
    if (dtype == DT_BF16) { if (nseq == 16) TT_DEC(bf16, 16); else TT_DEC(bf16, 4); }
    else { if (nseq == 16) TT_DEC(f16, 16); else TT_DEC(f16, 4); }
#undef TT_DEC
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }
  TT_REQUIRE(a.ngroups <= 1, "decode_attention: the prefixes of a multi-utterance batch (%d rows) do not fit the LDS", a.P1);
  const size_t smem = (size_t)4 * ctx_cap * sizeof(float);
  TT_REQUIRE(smem <= 64 * 1024, "decode_attention: context %d too long for the score buffer", ctx_cap);
  const int blocks = cdiv(a.B * a.heads, 4);
  if (dtype == DT_BF16) launch_timed(ps, decode_attn_kernel<bf16>, dim3(blocks), dim3(256), smem, stream, a, ctx_cap);
  else launch_timed(ps, decode_attn_kernel<f16>, dim3(blocks), dim3(256), smem, stream, a, ctx_cap);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
