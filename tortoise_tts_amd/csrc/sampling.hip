// On-device restatement of the HF transformers==4.31 sampling step that Tortoise's
// UnifiedVoice.inference_speech drives (reference: tortoise/models/autoregressive.py:535-563,
// loop skeleton tortoise/models/stream_generator.py:916-1000):
//   RepetitionPenalty(all input ids) -> /temperature -> top-k (ties kept) -> top-p (ascending
//   cumulative prob <= 1 - top_p removed, at least one kept) -> softmax -> multinomial
//   -> finished rows emit the stop token.
// One 256-thread block per candidate row, no host synchronisation: the step index lives in device
// memory so the whole decode step (this kernel included) replays from one hipGraph.
// multinomial(p, 1) is realised as argmax(p / q), q ~ Exp(1): q is either an injected tensor
// (parity runs share the oracle's draws) or Philox4x32-10 keyed by (seed; global row, step, token),
// which makes the sampled codes independent of how candidates are sharded over GPUs.
#include "ops.h"

namespace tt {

constexpr int SURV_CAP = 512;

// -DTT_SAMPLE_STAMPS (a variant build, scripts/sample_phases.py): thread 0 of block 0 files the 100 MHz wall clock at the phase boundaries
#ifdef TT_SAMPLE_STAMPS
__device__ unsigned long long g_sample_stamps[16];
#define TT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sample_stamps[i] = wall_clock64(); } while (0)
#else
#define TT_STAMP(i)
#endif

__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                               unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Strictly sequential (left to right) sum of x[0 .. n) in LDS, computed redundantly by every lane of ONE wave: the values are read 64
// at a time into a register per lane and handed out with v_readlane (a few cycles each) instead of one dependent LDS round trip
// (~100 cycles at one wave per SIMD) per element.  Same additions in the same order as `for (i) acc += x[i]`.
__device__ __forceinline__ float seq_sum_lds(const float* x, int n, int lane) {
  float acc = 0.f;
  for (int base = 0; base < n; base += 64) {
    const float v = base + lane < n ? x[base + lane] : 0.f;
    const int cnt = min(64, n - base);
    for (int j = 0; j < cnt; ++j) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
  }
  return acc;
}

// one thread: finished rows emit the stop token (stream_generator.py:980-996), bookkeeping of the sampled token; returns it
__device__ __forceinline__ int sample_commit(const SampleArgs& a, int b, int step, int best_i, unsigned* seen) {
  // a row of non-finite logits (an overflowed operand upstream: the guard has counted it) never beats the initial candidate: such a
  // row ends here with the stop token instead of indexing the tables with the sentinel
  if ((unsigned)best_i >= (unsigned)a.V) best_i = a.stop_token;
  const int unf = a.unfinished[b];
  const int tok = unf ? best_i : a.stop_token;
  const int still = unf && tok != a.stop_token;
  a.unfinished[b] = still;
  a.codes[(size_t)b * a.ldcodes + step] = tok;
  a.next_tok[b] = tok;
  atomicOr(&seen[tok >> 5], 1u << (tok & 31));
  if (still) atomicAdd(&a.unfinished_count[step], 1);
  return tok;
}
// whole block: next decode step's input row, mel_embedding[tok] + mel_pos_embedding[index of this token + offset].  The token
// sampled at the capacity limit is never fed back, and its position row would lie one past the table: skip it.
__device__ __forceinline__ void sample_embed(const SampleArgs& a, int b, int step, const int* tok_s, int tid, int nthreads) {
  if (a.embed_x && step + a.pos_offset < a.pos_len) {
    __syncthreads();
    const int tok = tok_s[0];
    const int pos = step + a.pos_offset;
    for (int c = tid * 4; c < a.D; c += nthreads * 4) {
      const float4 e = *(const float4*)(a.tok_emb + (size_t)tok * a.D + c);
      const float4 p = *(const float4*)(a.pos_emb + (size_t)pos * a.D + c);
      *(float4*)(a.embed_x + (size_t)b * a.D + c) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
  }
}

// PER: vocabulary entries per thread (V <= 256 PER): 33 covers the model's 8194 mel codes, 40 anything up to 10240 - every unrolled loop below is PER long
template <int PER>
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  __shared__ float cnt_s[2][3][4];
  __shared__ unsigned tmax_s[256];
  __shared__ unsigned long long pair_s[SURV_CAP];
  __shared__ unsigned wtot[4];
  __shared__ int nsurv, ncand;
  __shared__ float sv[SURV_CAP];
  __shared__ int si[SURV_CAP];
  __shared__ float sorted_v[SURV_CAP];
  __shared__ int sorted_i[SURV_CAP];
  __shared__ int kept;
  __shared__ float kept_total;
  __shared__ float red_v[4];
  __shared__ int red_i[4];

  const int b = blockIdx.x, tid = threadIdx.x;
  TT_STAMP(0);
  const int V = a.V;
  const int step = a.state[0];
  const int grp = a.ngroups > 1 ? b / a.group_size : 0;  // utterance of this row (block-uniform)
  // Philox key of this row's utterance: from device memory when the caller replays a cached step graph (the seed of a call is then
  // data, not a baked-in kernel argument), else from the argument block; block-uniform, read once
  const unsigned long long philox_key = a.keys_dev ? a.keys_dev[grp] : (a.ngroups > 1 ? a.group_seeds[grp] : a.seed);
  const int row_offset = a.row_offset_dev ? *a.row_offset_dev : a.row_offset;  // (block-uniform scalar load)
  const float* lg = a.logits + (a.ldl ? (size_t)b * a.ldl : (size_t)grp * (a.ldg ? a.ldg : V));
  unsigned* seen = a.seen + (size_t)b * ((V + 31) / 32);
  float val[PER];
  bool bad = false;  // NaN / +inf logits: an operand overflowed somewhere upstream (-inf is legitimate: a suppressed token)
  {
    // every load unconditional (index clamped) and requested before the first use: a `t < V` branch around the loads made the
    // compiler issue them one round trip at a time - 33 dependent global-memory latencies, 20 us of the 48 us this kernel took
    float raw[PER];
    unsigned sw[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int tc = min(tid + 256 * j, V - 1);
      raw[j] = lg[tc];
      sw[j] = seen[tc >> 5];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = tid + 256 * j;
      float s = raw[j];
      const bool live = t < V;
      bad = bad || (live && (s != s || s == INFINITY));
      if (a.rep_penalty != 1.0f && ((sw[j] >> (t & 31)) & 1u)) s = s < 0.f ? s * a.rep_penalty : s / a.rep_penalty;
      if (a.temperature != 1.0f) s = s / a.temperature;
      if (__float_as_uint(s) == 0x80000000u) s = 0.f;  // -0.0 ties with +0.0 in the reference's float comparisons: one zero, so that key order == float order
      val[j] = live ? s : -INFINITY;
    }
  }
  if (a.guard && __ballot(bad) != 0ull && (tid & 63) == 0) atomicAdd(a.guard, 1);
  TT_STAMP(1);

  // ---- top-k (ties kept) + order.  k (<= 256) is tiny against the vocabulary, so the work runs on a small candidate set:
  //   1. a lower bound L of the k-th largest key: the k-th largest of the 256 per-thread maxima, to 16 bits (k keys are >= it) -
  //      "greatest t with count(max >= t) >= k".  The maxima go through the LDS once; then every wave finds L on its own with ballot
  //      counts (4 maxima per lane, one bit per round, the count in scalar registers): one barrier instead of round 4's eight;
  //   2. every (score, token) with key >= L is a candidate (typically 60 - 120 of 8194): wave-aggregated compaction - ballot prefix
  //      inside the wave, ONE LDS atomic per wave for its base slot (the slot order is irrelevant: everything below is a total order);
  //   3. among the candidates: the k-th largest score = the top-k threshold (scores below it drop out, ties at it stay:
  //      TopKLogitsWarper) and every candidate's place in (score descending, token ascending) order, from ONE all-pairs count that
  //      all four waves share; a lane sees the other candidates through v_readlane of a register - round 4 read them one dependent
  //      LDS round trip at a time in two passes on one wave: 2.4 + 6.6 us of the kernel's 25 (scripts/sample_phases.py).
  // A plateau of equal scores that overflows the candidate buffer, or k beyond the populated threads, takes the generic path: a
  // counting search over all register-resident keys, 16 rounds.  (Round 3's radix select put ~8 000 LDS atomicAdds on three or
  // four bins in its first pass: logits share their sign / exponent byte.)
  const int k = a.top_k < V ? a.top_k : V;
  const int lane = tid & 63;
  unsigned key[PER];
  unsigned tmax = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    key[j] = tid + 256 * j < V ? f2key(val[j]) : 0u;  // (real keys are > 0: f2key(-inf) = 0x007FFFFF)
    tmax = max(tmax, key[j]);
  }
  tmax_s[tid] = tmax;
  pair_s[tid] = 0ull;
  pair_s[tid + 256] = 0ull;
  if (tid == 0) {
    nsurv = 0;
    ncand = 0;
  }
  __syncthreads();
  const float kf = (float)k;
  int round = 0;
  unsigned lower = 0u;
  {
    const unsigned m0 = tmax_s[lane], m1 = tmax_s[lane + 64], m2 = tmax_s[lane + 128], m3 = tmax_s[lane + 192];
#pragma unroll 1
    for (int bit = 31; bit >= 16; --bit) {
      const unsigned c = lower | (1u << bit);
      const int cnt = __popcll(__ballot(m0 >= c)) + __popcll(__ballot(m1 >= c)) + __popcll(__ballot(m2 >= c)) + __popcll(__ballot(m3 >= c));
      lower = cnt >= k ? c : lower;  // (wave-uniform)
    }
  }
  TT_STAMP(2);
  if (lower > 0u) {
    int wcnt = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) wcnt += __popcll(__ballot(key[j] >= lower));
    int off = 0;
    if (lane == 0) off = atomicAdd(&ncand, wcnt);
    off = __builtin_amdgcn_readfirstlane(off);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const bool c = key[j] >= lower;
      const unsigned long long m = __ballot(c);
      if (m != 0ull) {  // (wave-uniform)
        const int slot = off + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (c && slot < SURV_CAP) {
          sv[slot] = val[j];
          si[slot] = tid + 256 * j;
        }
        off += __popcll(m);
      }
    }
  }
  __syncthreads();
  const bool fast = lower > 0u && ncand <= SURV_CAP;  // (block-uniform)
  TT_STAMP(3);
#ifdef TT_SAMPLE_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_sample_stamps[15] = (unsigned long long)ncand;
#endif
  if (fast) {
    const int nc = __builtin_amdgcn_readfirstlane(ncand);
    // ONE all-pairs pass over the candidates: for candidate i, g = #{w_j > v_i}, eq = #{w_j == v_i}, lt = #{w_j == v_i, token_j < token_i}.
    // The k-th largest score is the candidate with g < k <= g + eq (ties kept: g + eq survivors), and i's place in (score descending,
    // token ascending) order is g + lt - the survivors are the first g + eq places, so no second pass separates them.  (Scores are
    // never -0.0 here, so the float order IS the key order.)  A lane owns candidate (64 m + lane) and reads the others 64 at a time
    // out of a register with v_readlane; the four waves split every 64 into 16s and add their partial counts into one packed LDS word
    // per candidate (16-bit fields: counts <= 512).
    const int wv = tid >> 6;
#pragma unroll 1
    for (int ibase = 0; ibase < nc; ibase += 64) {
      const int own = ibase + lane;
      const bool has = own < nc;
      const float v = has ? sv[own] : __builtin_nanf("");
      const int id = has ? si[own] : 0;
      int g = 0, eq = 0, lt = 0;
#pragma unroll 1
      for (int base = 0; base < nc; base += 64) {
        const bool in = base + lane < nc;
        const int wb = __builtin_bit_cast(int, in ? sv[base + lane] : __builtin_nanf(""));  // (NaN: counted by nobody)
        const int wid = in ? si[base + lane] : 0;
        const int jhi = min(wv * 16 + 16, nc - base);
        for (int j = wv * 16; j < jhi; ++j) {
          const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(wb, j));
          const int idj = __builtin_amdgcn_readlane(wid, j);
          g += wj > v ? 1 : 0;
          eq += wj == v ? 1 : 0;
          lt += (wj == v && idj < id) ? 1 : 0;
        }
      }
      if (has) atomicAdd(&pair_s[own], (unsigned long long)g | ((unsigned long long)eq << 16) | ((unsigned long long)lt << 32));
    }
    __syncthreads();
    TT_STAMP(4);
#pragma unroll 1
    for (int own = tid; own < nc; own += 256) {
      const unsigned long long pc = pair_s[own];
      const int g = (int)(pc & 0xFFFFu), eq = (int)((pc >> 16) & 0xFFFFu), lt = (int)((pc >> 32) & 0xFFFFu);
      sorted_v[g + lt] = sv[own];
      sorted_i[g + lt] = si[own];
      if (g < k && k <= g + eq) nsurv = g + eq;  // (every thread that qualifies writes the same count)
    }
    __syncthreads();
  } else {
    __syncthreads();
    if (tid == 0) nsurv = 0;
    unsigned thr = 0u;
#pragma unroll 1
    for (int bit = 30; bit >= 0; bit -= 2, ++round) {
      const unsigned c1 = thr | (1u << bit), c2 = thr | (2u << bit), c3 = thr | (3u << bit);
      float n1 = 0.f, n2 = 0.f, n3 = 0.f;  // counts <= 10 240: exact in fp32
#pragma unroll
      for (int j = 0; j < PER; ++j) {  // (fully unrolled: a run-time index into key[] would move the array to scratch memory)
        n1 += key[j] >= c1 ? 1.f : 0.f;
        n2 += key[j] >= c2 ? 1.f : 0.f;
        n3 += key[j] >= c3 ? 1.f : 0.f;
      }
      n1 = wave_sum(n1); n2 = wave_sum(n2); n3 = wave_sum(n3);
      float* cs = &cnt_s[round & 1][0][0];
      if ((tid & 63) == 0) {
        cs[0 * 4 + (tid >> 6)] = n1;
        cs[1 * 4 + (tid >> 6)] = n2;
        cs[2 * 4 + (tid >> 6)] = n3;
      }
      __syncthreads();
      const float t1 = cs[0] + cs[1] + cs[2] + cs[3], t2 = cs[4] + cs[5] + cs[6] + cs[7], t3 = cs[8] + cs[9] + cs[10] + cs[11];
      thr = t3 >= kf ? c3 : t2 >= kf ? c2 : t1 >= kf ? c1 : thr;
    }
    const unsigned kth = thr;  // key of the k-th largest score; everything >= kth survives (ties kept)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = tid + 256 * j;
      if (t < V && key[j] >= kth) {
        const int slot = atomicAdd(&nsurv, 1);
        if (slot < SURV_CAP) {
          sv[slot] = val[j];
          si[slot] = t;
        }
      }
    }
    __syncthreads();
    if (nsurv > SURV_CAP) {
      // Degenerate plateau: more than SURV_CAP scores tie at the k-th value (HF keeps them all).  The slots above were handed out in
      // atomic order, i.e. run-dependent: redo the selection deterministically - everything strictly above the k-th value (< k
      // entries), then the ties in ascending token order until the buffer is full.  (block-uniform branch: nsurv is shared)
      __syncthreads();
      if (tid == 0) nsurv = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int t = tid + 256 * j;
        if (t < V && key[j] > kth) {
          const int slot = atomicAdd(&nsurv, 1);
          sv[slot] = val[j];
          si[slot] = t;
        }
      }
      __syncthreads();
      int base = nsurv;
      // (fully unrolled although it is the rare path: a run-time index into val[] would move the whole array to scratch memory)
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (base < SURV_CAP) {  // block-uniform
          const int t = tid + 256 * j;
          const bool tie = t < V && key[j] == kth;
          const unsigned long long m = __ballot(tie);
          const int before = __popcll(m & ((1ull << (tid & 63)) - 1ull));
          if ((tid & 63) == 0) wtot[tid >> 6] = (unsigned)__popcll(m);
          __syncthreads();
          int off = base, total = 0;
          for (int w = 0; w < 4; ++w) {
            if (w < (tid >> 6)) off += (int)wtot[w];
            total += (int)wtot[w];
          }
          const int slot = off + before;
          if (tie && slot < SURV_CAP) {
            sv[slot] = val[j];
            si[slot] = t;
          }
          base += total;
          __syncthreads();
        }
      }
      if (tid == 0) nsurv = base;
      __syncthreads();
    }
    const int ns = nsurv < SURV_CAP ? nsurv : SURV_CAP;
    // rank sort: descending score, ascending index on ties (deterministic irrespective of slot order)
    for (int i = tid; i < ns; i += 256) {
      const float v = sv[i];
      const int id = si[i];
      int rank = 0;
#pragma unroll 8
      for (int j = 0; j < ns; ++j) {
        const float w = sv[j];
        rank += (w > v) || (w == v && si[j] < id);
      }
      sorted_v[rank] = v;
      sorted_i[rank] = id;
    }
    __syncthreads();
  }
  const int n = nsurv < SURV_CAP ? nsurv : SURV_CAP;
  TT_STAMP(5);
  // ---- top-p on the survivors (everything else already has probability 0) and the draw.  The three sums stay sequential in index
  // order so that the kept set is decided with exactly the arithmetic the oracle's cumulative sum uses.
  if (n <= 64) {
    // The usual case (k = 50; more than 64 survivors takes a tie plateau): the whole tail in wave 0 with the survivors in registers -
    // no LDS round trip, no barrier; every sum is the same left-to-right chain of additions as below, handed round with v_readlane
    // (lanes beyond the survivors hold +0.0, which changes no partial sum), fully unrolled: no loop or branch per element.
    if (tid < 64) {
      const bool in = tid < n;
      const float sl = in ? sorted_v[tid] : -INFINITY;
      const int id = in ? sorted_i[tid] : 0x7fffffff;
      const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl), 0));
      const float e = in ? __expf(sl - m) : 0.f;
      const int eb = __builtin_bit_cast(int, e);
      float total = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(eb, j));
      TT_STAMP(6);
      TT_STAMP(7);
      int keep = n;
      if (a.top_p < 1.0f) {
        // ascending cumulative probability of element r == sum of the probabilities of elements n-1 .. r; the partial sums never decrease,
        // so "the first r (from the top) whose sum exceeds 1 - top_p" is 1 + the number of r in [1, n) whose sum does
        const int pb = __builtin_bit_cast(int, e / total);
        const float thr = 1.0f - a.top_p;
        float tail = 0.f;
        int over = 0;
#pragma unroll
        for (int r = 63; r >= 1; --r) {
          tail += __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, r));
          over += tail > thr ? 1 : 0;
        }
        keep = over + 1;
      }
      const int kb = __builtin_bit_cast(int, tid < keep ? e : 0.f);
      float kt = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) kt += __builtin_bit_cast(float, __builtin_amdgcn_readlane(kb, j));
      TT_STAMP(8);
      // multinomial == argmax(p / q)
      float best = -1.f;
      int best_i = 0x7fffffff;
      if (tid < keep) {
        const float p = e / kt;
        float q;
        if (a.exp_noise) {
          q = a.exp_noise[((size_t)step * a.B + b) * V + id];
        } else {
          unsigned r[4];
          const unsigned long long key = philox_key;
          const int cand = a.ngroups > 1 ? b - grp * a.group_size : b;  // index within the utterance: the draw does not depend on the batching
          philox4x32_10((unsigned)id, (unsigned)step, (unsigned)(row_offset + cand), 0u, (unsigned)key, (unsigned)(key >> 32), r);
          const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
          q = -__logf(u);
        }
        const float sc = p / q;
        if (sc > best || (sc == best && id < best_i)) {
          best = sc;
          best_i = id;
        }
      }
      TT_STAMP(9);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) {
          best = ov;
          best_i = oi;
        }
      }
      if (tid == 0) red_i[0] = sample_commit(a, b, step, best_i, seen);
    }
  } else {
    // (a tie plateau: the exponentials by all threads through the LDS, the sums by wave 0)
    for (int i = tid; i < n; i += 256) sv[i] = __expf(sorted_v[i] - sorted_v[0]);  // sv is free after the rank sort
    __syncthreads();
    if (tid < 64) {  // (wave 0; every lane gets the same sum)
      const float total = seq_sum_lds(sv, n, tid);
      if (tid == 0) kept_total = total;  // (parked here for the parallel divisions below; overwritten with the kept sum afterwards)
    }
    __syncthreads();
    TT_STAMP(6);
    {  // the probabilities sv[r] / total, by all threads (the same IEEE division one thread did serially in round 3); si is free
      const float total = kept_total;
      float* pr = (float*)si;
      for (int i = tid; i < n; i += 256) pr[i] = sv[i] / total;
    }
    __syncthreads();
    TT_STAMP(7);
    if (tid < 64) {  // wave 0, every lane redundantly: the same sequential arithmetic as one thread walking the arrays
      const float* pr = (const float*)si;
      int keep = n;
      if (a.top_p < 1.0f) {
        float tail = 0.f;
        keep = 1;
        const float thr = 1.0f - a.top_p;
        // ascending cumulative probability of element r == sum of probabilities of elements r..n-1
        bool done = false;
        for (int hi = n - 1; hi >= 1 && !done; hi -= 64) {  // elements hi, hi - 1, ... in chunks of 64 (lane j holds element hi - j)
          const int r_l = hi - tid;
          const float v = r_l >= 1 ? pr[r_l] : 0.f;
          const int cnt = min(64, hi);
          for (int j = 0; j < cnt; ++j) {
            tail += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
            if (tail > thr) {
              keep = hi - j + 1;
              done = true;
              break;
            }
          }
        }
      }
      const float kt = seq_sum_lds(sv, keep, tid);
      if (tid == 0) {
        kept = keep;
        kept_total = kt;
      }
    }
    __syncthreads();
    TT_STAMP(8);
    // ---- multinomial == argmax(p / q)
    const int keep = kept;
    const float m = sorted_v[0];
    float best = -1.f;
    int best_i = 0x7fffffff;
    for (int i = tid; i < keep; i += 256) {
      const int id = sorted_i[i];
      const float p = __expf(sorted_v[i] - m) / kept_total;
      float q;
      if (a.exp_noise) {
        q = a.exp_noise[((size_t)step * a.B + b) * V + id];
      } else {
        unsigned r[4];
        const unsigned long long key = philox_key;
        const int cand = a.ngroups > 1 ? b - grp * a.group_size : b;  // index within the utterance: the draw does not depend on the batching
        philox4x32_10((unsigned)id, (unsigned)step, (unsigned)(row_offset + cand), 0u, (unsigned)key, (unsigned)(key >> 32), r);
        const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
        q = -__logf(u);
      }
      const float sc = p / q;
      if (sc > best || (sc == best && id < best_i)) {
        best = sc;
        best_i = id;
      }
    }
    TT_STAMP(9);
    // block argmax (value desc, index asc)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(best_i, o, 64);
      if (ov > best || (ov == best && oi < best_i)) {
        best = ov;
        best_i = oi;
      }
    }
    if ((tid & 63) == 0) {
      red_v[tid >> 6] = best;
      red_i[tid >> 6] = best_i;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (red_v[w] > best || (red_v[w] == best && red_i[w] < best_i)) {
          best = red_v[w];
          best_i = red_i[w];
        }
      red_i[0] = sample_commit(a, b, step, best_i, seen);
    }
  }
  TT_STAMP(10);
  sample_embed(a, b, step, red_i, tid, 256);
  TT_STAMP(11);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Any top_k (HF accepts every value: TopKLogitsWarper clamps k to the vocabulary, top_k == 0 drops the warper; reference loop
// stream_generator.py:43-57 builds the same warper list).  The kernel above keeps its <= 512 survivors in a rank-sorted buffer,
// which is what the reference's own default (k = 50) and anything up to 256 need; beyond that the whole row is sorted: one
// 1024-thread workgroup per candidate, bitonic sort of (score key, token) pairs in LDS (16384 slots: 96 KB + 40 KB of exponentials),
// descending score / ascending token on ties, then exactly the same top-p arithmetic and argmax(p / q) draw on the first n entries.
// A correct path for every k, not a fast one: the reference's own default (k = 50) and the benchmark take the kernel above.
constexpr int WIDE_N = 16384;
constexpr int WIDE_V = 10240;
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(1024) void sample_wide_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wide[];
  unsigned* key = (unsigned*)smem_wide;                    // [WIDE_N]
  unsigned short* tok = (unsigned short*)(key + WIDE_N);   // [WIDE_N]
  float* ef = (float*)(tok + WIDE_N);                      // [WIDE_V]
  __shared__ int n_s, kept;
  __shared__ float kept_total;
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = a.V;
  const int step = a.state[0];
  const int grp = a.ngroups > 1 ? b / a.group_size : 0;
  // Philox key of this row's utterance: from device memory when the caller replays a cached step graph (the seed of a call is then
  // data, not a baked-in kernel argument), else from the argument block; block-uniform, read once
  const unsigned long long philox_key = a.keys_dev ? a.keys_dev[grp] : (a.ngroups > 1 ? a.group_seeds[grp] : a.seed);
  const int row_offset = a.row_offset_dev ? *a.row_offset_dev : a.row_offset;
  const float* lg = a.logits + (a.ldl ? (size_t)b * a.ldl : (size_t)grp * (a.ldg ? a.ldg : V));
  unsigned* seen = a.seen + (size_t)b * ((V + 31) / 32);
  for (int t = tid; t < WIDE_N; t += 1024) {
    unsigned kk = 0u;  // padding slots sort behind every real score (f2key(-inf) = 0x007FFFFF > 0)
    if (t < V) {
      float s = lg[t];
      if (a.rep_penalty != 1.0f && ((seen[t >> 5] >> (t & 31)) & 1u)) s = s < 0.f ? s * a.rep_penalty : s / a.rep_penalty;
      if (a.temperature != 1.0f) s = s / a.temperature;
      kk = f2key(s);
    }
    key[t] = kk;
    tok[t] = (unsigned short)(t < V ? t : 0xFFFF);
  }
  // bitonic sort: final order = descending key, ascending token on equal keys
  for (int k2 = 2; k2 <= WIDE_N; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int p = tid; p < WIDE_N / 2; p += 1024) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
        const unsigned ki = key[i], kl = key[l];
        const unsigned short ti = tok[i], tl = tok[l];
        const bool i_first = ki > kl || (ki == kl && ti < tl);  // i already precedes l in the final order
        const bool up = (i & k2) == 0;                           // this sub-sequence is built in final order (else reversed)
        if (up != i_first) {
          key[i] = kl; key[l] = ki;
          tok[i] = tl; tok[l] = ti;
        }
      }
    }
  }
  __syncthreads();
  // survivors of top-k: everything >= the k-th score (ties kept, like TopKLogitsWarper's `scores < kth` removal)
  const int k_eff = (a.top_k <= 0 || a.top_k > V) ? V : a.top_k;
  const unsigned kth = key[k_eff - 1];
  for (int t = tid; t < V; t += 1024)
    if (key[t] >= kth && (t + 1 == V || key[t + 1] < kth)) n_s = t + 1;
  __syncthreads();
  const int n = n_s;
  const float v0 = key2f(key[0]);
  for (int i = tid; i < n; i += 1024) ef[i] = __expf(key2f(key[i]) - v0);
  __syncthreads();
  if (tid == 0) {  // same sequential arithmetic as sample_kernel
    float total = 0.f;
    for (int i = 0; i < n; ++i) total += ef[i];
    int keep = n;
    if (a.top_p < 1.0f) {
      float tail = 0.f;
      keep = 1;
      const float thr = 1.0f - a.top_p;
      for (int r = n - 1; r >= 1; --r) {
        tail += ef[r] / total;
        if (tail > thr) {
          keep = r + 1;
          break;
        }
      }
    }
    float kt = 0.f;
    for (int i = 0; i < keep; ++i) kt += ef[i];
    kept = keep;
    kept_total = kt;
  }
  __syncthreads();
  const int keep = kept;
  float best = -1.f;
  int best_i = 0x7fffffff;
  for (int i = tid; i < keep; i += 1024) {
    const int id = tok[i];
    const float p = __expf(key2f(key[i]) - v0) / kept_total;
    float q;
    if (a.exp_noise) {
      q = a.exp_noise[((size_t)step * a.B + b) * V + id];
    } else {
      unsigned r[4];
      const unsigned long long pk = philox_key;
      const int cand = a.ngroups > 1 ? b - grp * a.group_size : b;
      philox4x32_10((unsigned)id, (unsigned)step, (unsigned)(row_offset + cand), 0u, (unsigned)pk, (unsigned)(pk >> 32), r);
      const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      q = -__logf(u);
    }
    const float sc = p / q;
    if (sc > best || (sc == best && id < best_i)) {
      best = sc;
      best_i = id;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ov > best || (ov == best && oi < best_i)) {
      best = ov;
      best_i = oi;
    }
  }
  if ((tid & 63) == 0) {
    red_v[tid >> 6] = best;
    red_i[tid >> 6] = best_i;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (red_v[w] > best || (red_v[w] == best && red_i[w] < best_i)) {
        best = red_v[w];
        best_i = red_i[w];
      }
    red_i[0] = sample_commit(a, b, step, best_i, seen);
  }
  sample_embed(a, b, step, red_i, tid, 1024);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Typical sampling: the reference's TypicalLogitsWarper (tortoise/utils/typical_sampling.py:11-33), which inference_speech puts into
// generate()'s logits_processor list when tts() is called with typical_sampling=True (api.py:361-364, autoregressive.py:558) - it runs
// AFTER the repetition penalty and BEFORE temperature / top-k / top-p.  Per row, on the penalised scores s:
//   logp = log_softmax(s), p = exp(logp), H = -nansum(logp p), d = |(-logp) - H|;
//   tokens in ascending order of d, c_j = cumsum(softmax(s in that order)); last = #{j : c_j < mass}; everything with d > d_(last) is removed.
// c is nondecreasing, so d_(last) = tau = the smallest distance whose closed set {d <= tau} carries a probability >= mass - no sort is
// needed: tau is found bit by bit over the (non-negative) float keys of d, 31 counting rounds, each one block-wide sum.  The sums run
// in double (the reference's CPU cumsum accumulates float probabilities in double and compares the float-rounded value with mass),
// partials combined in a fixed order: the kept set does not depend on the launch geometry.  One 1024-thread workgroup per logits row.
// The kernel only MASKS: kept tokens keep their RAW logit (the sampler applies the penalty itself; a removed token stays -inf under it).
constexpr int TYP_PER = WIDE_V / 1024;

__device__ __forceinline__ double typ_block_sum(double v, double* slots /* [16] of this round */, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((tid & 63) == 0) slots[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += slots[w];
  return t;
}

__global__ __launch_bounds__(1024) void typical_mask_kernel(SampleArgs a) {
  __shared__ double red_d[4][16];
  __shared__ float red_f[16];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int V = a.V;
  // row r of the logits: a candidate (ldl != 0) or, while every candidate of a group still shares the prefill logits, the group's row;
  // the ids seen so far are then the same for all its candidates (the fake prefix ids): the first one's mask is read
  const size_t row_off = a.ldl ? (size_t)r * a.ldl : (size_t)r * (a.ldg ? a.ldg : V);
  const int seen_row = a.ldl ? r : (a.ngroups > 1 ? r * a.group_size : 0);
  const float* lg = a.logits + row_off;
  float* out = a.typical_out + row_off;
  const unsigned* seen = a.seen + (size_t)seen_row * ((V + 31) / 32);
  float raw[TYP_PER], s[TYP_PER];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < TYP_PER; ++j) {
    const int t = tid + 1024 * j;
    const int tc = min(t, V - 1);
    raw[j] = lg[tc];
    float x = raw[j];
    if (a.rep_penalty != 1.0f && ((seen[tc >> 5] >> (tc & 31)) & 1u)) x = x < 0.f ? x * a.rep_penalty : x / a.rep_penalty;
    s[j] = t < V ? x : -INFINITY;
    m = fmaxf(m, s[j]);
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red_f[tid >> 6] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) m = fmaxf(m, red_f[w]);
  // softmax / log_softmax pieces as ATen forms them: e = exp(s - max), total = sum e, logp = (s - max) - log(total)
  float e[TYP_PER];
  double part = 0.0;
#pragma unroll
  for (int j = 0; j < TYP_PER; ++j) {
    e[j] = expf(s[j] - m);  // exp(-inf) = 0: suppressed tokens and the padding beyond V
    part += (double)e[j];
  }
  const float total = (float)typ_block_sum(part, red_d[0], tid);
  const float log_total = logf(total);
  float logp[TYP_PER];
  part = 0.0;
#pragma unroll
  for (int j = 0; j < TYP_PER; ++j) {
    logp[j] = (s[j] - m) - log_total;
    const float prod = logp[j] * expf(logp[j]);
    if (prod == prod) part += (double)prod;  // nansum: -inf * 0 of a suppressed token does not count
  }
  const float ent = -(float)typ_block_sum(part, red_d[1], tid);
  unsigned key[TYP_PER];
  float pc[TYP_PER];
#pragma unroll
  for (int j = 0; j < TYP_PER; ++j) {
    key[j] = tid + 1024 * j < V ? (__float_as_uint(fabsf((-logp[j]) - ent)) & 0x7FFFFFFFu) : 0x7FFFFFFFu;
    pc[j] = e[j] / total;
  }
  // tau = the greatest c with (float)sum{pc : key < c} < mass  ==  the smallest c whose closed set reaches the mass
  unsigned tau = 0u;
#pragma unroll 1
  for (int bit = 30; bit >= 0; --bit) {
    const unsigned c = tau | (1u << bit);
    part = 0.0;
#pragma unroll
    for (int j = 0; j < TYP_PER; ++j) part += key[j] < c ? (double)pc[j] : 0.0;
    const float below = (float)typ_block_sum(part, red_d[2 + (bit & 1)], tid);  // two slot sets: one barrier per round
    tau = below < a.typical_mass ? c : tau;  // (block-uniform: every thread adds the same 16 partials in the same order)
  }
#pragma unroll
  for (int j = 0; j < TYP_PER; ++j) {
    const int t = tid + 1024 * j;
    if (t < V) out[t] = key[j] <= tau ? raw[j] : -INFINITY;
  }
}

int typical_mask_launch(const SampleArgs& a, hipStream_t stream) {
  TT_REQUIRE(a.B > 0 && a.V > 0 && a.V <= WIDE_V, "typical mask: V=%d unsupported (<= %d)", a.V, WIDE_V);
  TT_REQUIRE(a.typical_mass > 0.f && a.typical_mass < 1.f, "sample: typical_mass %g outside (0, 1)", (double)a.typical_mass);
  TT_REQUIRE(a.typical_out != nullptr && a.rep_penalty > 0.f, "sample: typical sampling needs a row buffer and a positive repetition penalty");
  const int rows = a.ldl ? a.B : (a.ngroups > 1 ? a.ngroups : 1);
  hipLaunchKernelGGL(typical_mask_kernel, dim3(rows), dim3(1024), 0, stream, a);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

int sample_launch(const SampleArgs& a_in, hipStream_t stream) {
  SampleArgs a = a_in;
  TT_REQUIRE(a.B > 0 && a.V > 0 && a.V <= WIDE_V, "sample: V=%d unsupported (<= %d)", a.V, WIDE_V);
  TT_REQUIRE(a.ngroups <= 1 || (a.ngroups <= 16 && a.group_size > 0 && a.B == a.ngroups * a.group_size), "sample: %d groups of %d rows do not make %d rows", a.ngroups, a.group_size, a.B);
  TT_REQUIRE(a.temperature > 0.f && a.top_p > 0.f && a.rep_penalty > 0.f, "sample: bad sampling parameters");
  if (a.typical_mass != 0.f) {
    TT_TRY(typical_mask_launch(a, stream));
    a.logits = a.typical_out;
  }
  ProfScope ps(PROF_SAMPLE, stream, 0.0, (double)a.B * a.V * 4.0, true);
  if (a.top_k >= 1 && a.top_k <= 256) {
    if (a.V <= 33 * 256) launch_timed(ps, sample_kernel<33>, dim3(a.B), dim3(256), 0, stream, a);
    else launch_timed(ps, sample_kernel<40>, dim3(a.B), dim3(256), 0, stream, a);
  } else {  // top_k == 0 (HF: no top-k warper), > 256, or beyond the vocabulary: the full-sort kernel
    constexpr size_t smem = (size_t)WIDE_N * 6 + (size_t)WIDE_V * 4;
    static bool attr_done = false;
    if (!attr_done) {
      TT_CHECK_HIP(hipFuncSetAttribute((const void*)sample_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      attr_done = true;
    }
    launch_timed(ps, sample_wide_kernel, dim3(a.B), dim3(1024), smem, stream, a);
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef TT_SAMPLE_STAMPS
}  // namespace tt
extern "C" int ttx_sample_stamps(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(tt::g_sample_stamps), sizeof(tt::g_sample_stamps));
}
namespace tt {
#endif

// Last kernel of a step.  state[0] = tokens sampled so far, state[1] = slot / index of the newest token (the one the next decode
// step feeds), state[2] = index of the first token after which every row had stopped (-1: none yet).  With `progress` (pinned
// host memory) the two host-visible words are published with system-scope stores: the host paces and ends its launch loop on
// them without draining the queue (tt_ar_generate).
__global__ void ar_advance_kernel(int* state, const int* unfinished_count, int* progress) {
  if (threadIdx.x == 0) {
    const int step = state[0];  // index of the token the sampler just produced
    state[0] = step + 1;
    state[1] = step;
    if (progress) {
      if (unfinished_count && unfinished_count[step] == 0 && state[2] < 0) {
        state[2] = step;
        __hip_atomic_store(progress + 1, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(progress, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
int ar_state_advance_launch(int* state, const int* unfinished_count, int* progress, hipStream_t stream) {
  ar_advance_kernel<<<1, 64, 0, stream>>>(state, unfinished_count, progress);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void ar_begin_kernel(int* state, unsigned* seen, int* unfinished, int* unfinished_count, int B, int V, int max_steps,
                                int start_token) {
  const int words = (V + 31) / 32;
  const int total = B * words;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int w = i % words;
    unsigned v = 0u;
    // fake_inputs = [1] * P + [start_mel_token] (autoregressive.py:546-548): ids 1 and start are "seen"
    if (w == 0) v |= 1u << 1;
    if (w == (start_token >> 5)) v |= 1u << (start_token & 31);
    seen[i] = v;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) unfinished[i] = 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max_steps; i += gridDim.x * blockDim.x) unfinished_count[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state[0] = 0;
    state[1] = -1;
    state[2] = -1;
  }
}
int ar_begin_launch(int* state, unsigned* seen, int* unfinished, int* unfinished_count, int B, int V, int max_steps,
                    int start_token, hipStream_t stream) {
  ar_begin_kernel<<<64, 256, 0, stream>>>(state, seen, unfinished, unfinished_count, B, V, max_steps, start_token);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void ar_embed_kernel(const int* tok, const int* state, const float* tok_emb, const float* pos_emb,
                                                       float* x, int D, int pos_offset) {
  const int b = blockIdx.x;
  const int t = tok[b];
  const int pos = state[1] + pos_offset;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    const float4 e = *(const float4*)(tok_emb + (size_t)t * D + c);
    const float4 p = *(const float4*)(pos_emb + (size_t)pos * D + c);
    *(float4*)(x + (size_t)b * D + c) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
  }
}
int ar_embed_launch(const int* tok, const int* state, const float* tok_emb, const float* pos_emb, float* x, int B, int D, int pos_offset,
                    hipStream_t stream) {
  TT_REQUIRE(D % 4 == 0, "ar_embed: D must be a multiple of 4");
  ar_embed_kernel<<<B, 256, 0, stream>>>(tok, state, tok_emb, pos_emb, x, D, pos_offset);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace tt
