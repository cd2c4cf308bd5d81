// Kernel microbenchmarks and A/B experiments (NOT part of the product library).  Built into
// tortoise_tts_amd/lib/libtortoise_kbench.so by `python -m tortoise_tts_amd.build --kbench` together with the product
// sources, and driven by scripts/kbench.py on the MI355X.  Every case captures a chain of launches into a hipGraph and
// replays it, so the reported time is device time per launch, not host launch rate.
#include "../runtime.h"
#include <vector>
#include <string>

using namespace tt;

namespace {

struct GraphTimer {
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  int init() {
    TT_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    TT_CHECK_HIP(hipEventCreate(&a));
    TT_CHECK_HIP(hipEventCreate(&b));
    return 0;
  }
  // enqueue(stream) is captured once; the graph is replayed `reps` times; returns microseconds per replay
  template <typename F> int run(F enqueue, int reps, double* us_out) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    TT_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = enqueue(s);
    hipError_t ce = hipStreamEndCapture(s, &g);
    if (rc) return rc;
    TT_CHECK_HIP(ce);
    TT_CHECK_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 2; ++i) TT_CHECK_HIP(hipGraphLaunch(ex, s));
    TT_CHECK_HIP(hipStreamSynchronize(s));
    TT_CHECK_HIP(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) TT_CHECK_HIP(hipGraphLaunch(ex, s));
    TT_CHECK_HIP(hipEventRecord(b, s));
    TT_CHECK_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    TT_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
    *us_out = 1e3 * ms / reps;
    (void)hipGraphExecDestroy(ex);
    (void)hipGraphDestroy(g);
    return 0;
  }
  void destroy() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (s) (void)hipStreamDestroy(s);
  }
};

__global__ void fill_kernel(unsigned short* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float f = ((float)(x & 0xFFFF) / 65536.f - 0.5f) * 0.25f;  // full-range signs (DVFS: never bench on zeros)
    __bf16 h = (__bf16)f;
    p[i] = *(unsigned short*)&h;
  }
}
__global__ void fill_f32_kernel(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((float)(x & 0xFFFF) / 65536.f - 0.5f);
  }
}
static int dev_bf16(Arena& ar, void** p, size_t n, unsigned seed) {
  TT_TRY(ar.alloc(p, n * 2, false));
  fill_kernel<<<1024, 256>>>((unsigned short*)*p, n, seed);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}
static int dev_f32(Arena& ar, float** p, size_t n, unsigned seed) {
  TT_TRY(ar.alloc((void**)p, n * 4, false));
  fill_f32_kernel<<<1024, 256>>>(*p, n, seed);
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// --------------------------------------------------------------------------------------------------------------------
// Experimental GEMM: same data path as the product's direct-to-LDS kernel with the degrees of freedom exposed:
// wave grid WM x WN over the BM x BN tile, ring depth ST, min waves per SIMD (blocks per CU), and ablation modes
//   MODE 0 full | 1 no loads inside the k-loop (LDS + MFMA + barrier floor) | 2 no LDS reads / MFMA (memory pipeline floor).
// out[m][n] (bf16) = sum_k A[m][k] W[n][k].  Optional 3-tap conv addressing (CONV) like the product kernel.
__device__ __attribute__((aligned(16))) unsigned int kb_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
typedef __attribute__((address_space(3))) void lds_void_k;
typedef __attribute__((address_space(1))) const void gbl_void_k;

struct ExpArgs {
  const bf16* A; const bf16* W; bf16* out; int M, N, K, lda, ldw, ldo, taps, seq_len, cin;
  const float* bias;  // MODE 3: f32 bias quads requested behind the ring fill, added in the epilogue
};

template <int BM, int BN, int WM, int WN, int ST, int MODE, int MINW, bool CONV>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_exp_kernel(ExpArgs g) {
  typedef Vec<bf16>::x8 x8;
  constexpr int NW = WM * WN, BK = 64;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;
  static_assert(PA >= 1 && PW >= 1 && FM >= 1 && FN >= 1, "bad tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16* As = (bf16*)smem_raw;
  bf16* Ws = As + ST * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  int bx, by;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y, id = blockIdx.x + gx * blockIdx.y;
    const int xcd = id & 7, loc = id >> 3, q = nwg >> 3, r = nwg & 7;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    bx = nid % gx; by = nid / gx;
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int nk = g.K / BK;
  const bf16* zero = (const bf16*)kb_zero_page;
  const int lr = lane >> 3, lc = lane & 7;
  int a_b[PA], a_s[PA], a_src[PA];
  bool a_ok[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = (wave + NW * p) * 8 + lr;
    a_src[p] = (lc ^ ((row >> 1) & 7)) * 8;
    const int m = m0 + row;
    a_ok[p] = m < g.M;
    if (CONV) { a_b[p] = m / g.seq_len; a_s[p] = m - a_b[p] * g.seq_len; }
    else { a_b[p] = 0; a_s[p] = a_ok[p] ? m : 0; }
  }
  const bf16* w_ptr[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int row = (wave + NW * p) * 8 + lr;
    const int n = n0 + row;
    w_ptr[p] = g.W + (size_t)(n < g.N ? n : g.N - 1) * g.ldw + (lc ^ ((row >> 1) & 7)) * 8;
  }
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
    int tap = 0, kin = k0;
    if (CONV) { tap = k0 / g.cin; kin = k0 - tap * g.cin; }
    const int shift = tap - (g.taps >> 1);
    bf16* as = As + buf * BM * BK;
    bf16* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const bf16* src;
      if (CONV) {
        const int s2 = a_s[p] + shift;
        const bool ok = a_ok[p] && s2 >= 0 && s2 < g.seq_len;
        src = ok ? g.A + ((size_t)a_b[p] * g.seq_len + s2) * g.lda + kin + a_src[p] : zero;
      } else {
        src = g.A + (size_t)a_s[p] * g.lda + kin + a_src[p];
      }
      __builtin_amdgcn_global_load_lds((gbl_void_k*)src, (lds_void_k*)(as + (wave + NW * p) * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void_k*)(w_ptr[p] + (size_t)kt * BK), (lds_void_k*)(ws + (wave + NW * p) * 8 * BK), 16, 0, 0);
  };
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int buf) {
    const bf16* as = As + buf * BM * BK;
    const bf16* ws = Ws + buf * BN * BK;
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
  constexpr int G = PA + PW;
  const int last = nk - 1;
#pragma unroll
  for (int s = 0; s < ST - 1; ++s) issue(min(s, last), s);
  float4 bq[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) bq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 3) {
#pragma unroll
    for (int i = 0; i < FN; ++i) bq[i] = *(const float4*)(g.bias + min(n0 + wn * TN + i * 16 + fg * 4, g.N - 4));
  }
  int slot = 0;
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int kt = 0; kt < nk; ++kt) {
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * G) : "memory");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + ST - 1;
    if (nslot >= ST) nslot -= ST;
    if (MODE != 1) issue(min(kt + ST - 1, last), nslot);
    if (MODE != 2) compute(slot);
    slot = slot + 1 == ST ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // epilogue: bf16 stores, 8 B per lane
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * TM + j * 16 + fr, n = n0 + wn * TN + i * 16 + fg * 4;
      if (m < g.M && n < g.N) *(Vec<bf16>::x4*)(g.out + (size_t)m * g.ldo + n) = pack4<bf16>(acc[i][j][0] + bq[i].x, acc[i][j][1] + bq[i].y, acc[i][j][2] + bq[i].z, acc[i][j][3] + bq[i].w);
    }
}

template <int BM, int BN, int WM, int WN, int ST, int MODE, int MINW>
static int launch_exp(const ExpArgs& a, hipStream_t s) {
  constexpr int smem = ST * (BM + BN) * 64 * 2;
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN));
  if (a.taps > 1) {
    auto fn = gemm_exp_kernel<BM, BN, WM, WN, ST, MODE, MINW, true>;
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    fn<<<grid, WM * WN * 64, smem, s>>>(a);
  } else {
    auto fn = gemm_exp_kernel<BM, BN, WM, WN, ST, MODE, MINW, false>;
    TT_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    fn<<<grid, WM * WN * 64, smem, s>>>(a);
  }
  TT_CHECK_HIP(hipGetLastError());
  return 0;
}

// variant ids: 100*config + mode
//   config 0: 128x64 8 waves 2x4 ring 4 (the product's denoiser tile)      1: 128x64 4 waves 2x2 ring 4
//          2: 128x64 4 waves 2x2 ring 3, 2 blocks/CU                        3: 128x64 8 waves 4x2 ring 4
//          4: 64x64 4 waves 2x2 ring 4 (the product's decode tile)          5: 64x64 4 waves ring 3, 2 blocks/CU
//          6: 128x128 8 waves 2x4 ring 3                                    7: 128x128 4 waves 2x2 ring 3
//          8: 256x64 8 waves 4x2 ring 3                                     9: 128x64 8 waves 2x4 ring 6
static int launch_variant(int variant, const ExpArgs& a, hipStream_t s) {
  switch (variant) {
#define V3(cfg, ...) \
    case cfg * 100 + 0: return launch_exp<__VA_ARGS__, 0, 1>(a, s); \
    case cfg * 100 + 1: return launch_exp<__VA_ARGS__, 1, 1>(a, s); \
    case cfg * 100 + 2: return launch_exp<__VA_ARGS__, 2, 1>(a, s); \
    case cfg * 100 + 3: return launch_exp<__VA_ARGS__, 3, 1>(a, s);
    V3(0, 128, 64, 2, 4, 4)
    V3(1, 128, 64, 2, 2, 4)
    V3(3, 128, 64, 4, 2, 4)
    V3(4, 64, 64, 2, 2, 4)
    V3(6, 128, 128, 2, 4, 3)
    V3(7, 128, 128, 2, 2, 3)
    V3(8, 256, 64, 4, 2, 3)
    V3(9, 128, 64, 2, 4, 6)
    // decode-shape ring-depth series: how much of the per-launch time is "bytes in flight" 
    V3(30, 64, 64, 2, 2, 8)     // 64x64 4 waves, 8-stage ring (128 KB LDS, 7 tiles = 112 KB in flight)
    V3(31, 64, 64, 2, 2, 10)    // 10-stage ring (160 KB LDS = the whole CU)
    V3(32, 32, 32, 2, 2, 16)    // 32x32 4 waves, 16 stages of 8 KB: ALL of K = 1024 in flight at once (no-slab decode projection)
    V3(33, 32, 32, 2, 2, 20)    // 32x32, 20 stages (160 KB): K = 4096 streamed through
    V3(34, 64, 64, 4, 2, 10)    // 64x64 8 waves (16x32 per wave), 10 stages
    V3(35, 64, 64, 2, 2, 6)     // 6-stage ring
    V3(36, 32, 64, 2, 2, 12)    // 32x64 4 waves (16x32 per wave), 12 stages of 12 KB
    case 200: return launch_exp<128, 64, 2, 2, 3, 0, 2>(a, s);
    case 500: return launch_exp<64, 64, 2, 2, 3, 0, 2>(a, s);
    case 501: return launch_exp<64, 64, 2, 2, 2, 0, 4>(a, s);   // 4 blocks / CU (16 KB LDS each x 2 stages)
    case 502: return launch_exp<64, 64, 2, 2, 3, 0, 3>(a, s);   // 3 blocks / CU
    case 1300: return launch_exp<128, 64, 2, 4, 3, 0, 2>(a, s);  // 8 waves, 2 blocks / CU (72 KB LDS each)
    case 1301: return launch_exp<128, 64, 4, 2, 3, 0, 2>(a, s);
    case 1600: return launch_exp<128, 128, 2, 4, 2, 0, 2>(a, s); // the product's large-M tile: 2 stages, 2 blocks / CU
    // large-M candidates (CLVP, conditioning-integrator pre-pass, a batched denoiser): more flops per byte through the per-CU load path
    case 2000: return launch_exp<256, 128, 4, 2, 3, 0, 1>(a, s);  // 8 waves (64x64 per wave), 3 stages = 144 KB
    case 2100: return launch_exp<256, 128, 4, 4, 3, 0, 1>(a, s);  // 16 waves (64x32 per wave), 3 stages
    case 2200: return launch_exp<256, 256, 4, 4, 2, 0, 1>(a, s);  // 16 waves (64x64 per wave), 2 stages = 128 KB
    case 2300: return launch_exp<256, 256, 2, 4, 2, 0, 1>(a, s);  // 8 waves (128x64 per wave), 2 stages
    case 2400: return launch_exp<128, 256, 2, 4, 3, 0, 1>(a, s);  // 8 waves (64x64 per wave), 3 stages, W-heavy tile
#undef V3
  }
  set_error("kbench: unknown gemm variant %d", variant);
  return -1;
}

extern "C" {

// One GEMM shape, `nw` distinct weight matrices visited round-robin (nw large => HBM-cold weights), `chain` launches per
// graph.  pad: extra elements per A / W row (row strides K + pad: spreads rows over L2 channels).  us_out = microseconds per launch.
int tt_kb_gemm_exp(int variant, int M, int N, int K, int taps, int seq_len, int nw, int chain, int pad, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  void* A = nullptr; void* out = nullptr;
  std::vector<void*> W(nw);
  const int lda = K / taps + pad, ldw = K + pad;
  float* bias = nullptr;
  int rc = dev_bf16(ar, &A, (size_t)(M + 8) * lda, 1u);
  if (!rc) rc = ar.alloc(&out, (size_t)M * N * 2);
  if (!rc) rc = dev_f32(ar, &bias, N + 64, 5u);
  for (int i = 0; i < nw && !rc; ++i) rc = dev_bf16(ar, &W[i], (size_t)N * ldw, 77u + i);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      ExpArgs a;
      memset(&a, 0, sizeof(a));
      a.A = (const bf16*)A; a.W = (const bf16*)W[i % nw]; a.out = (bf16*)out; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = N;
      a.taps = taps; a.seq_len = seq_len > 0 ? seq_len : M; a.cin = K / taps; a.bias = bias;
      TT_TRY(launch_variant(variant, a, s));
    }
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

// Same as tt_kb_gemm_exp with `na` distinct activation matrices visited round-robin as well (na > 1: the A operand is not
// L2-resident from the previous launch - the in-situ situation of the decode step, where another kernel has just produced it).
int tt_kb_gemm_exp_na(int variant, int M, int N, int K, int nw, int na, int chain, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  void* out = nullptr;
  std::vector<void*> W(nw), Av(na);
  float* bias = nullptr;
  int rc = 0;
  for (int i = 0; i < na && !rc; ++i) rc = dev_bf16(ar, &Av[i], (size_t)(M + 8) * K, 1u + i);
  if (!rc) rc = ar.alloc(&out, (size_t)M * N * 2);
  if (!rc) rc = dev_f32(ar, &bias, N + 64, 5u);
  for (int i = 0; i < nw && !rc; ++i) rc = dev_bf16(ar, &W[i], (size_t)N * K, 77u + i);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      ExpArgs a;
      memset(&a, 0, sizeof(a));
      a.A = (const bf16*)Av[i % na]; a.W = (const bf16*)W[i % nw]; a.out = (bf16*)out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldo = N;
      a.taps = 1; a.seq_len = M; a.cin = K; a.bias = bias;
      TT_TRY(launch_variant(variant, a, s));
    }
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

// The product GEMM (gemm_launch) on the same harness: one shape, nw weight copies, optional split-K, f32 or T output.
int tt_kb_gemm_prod(int M, int N, int K, int taps, int seq_len, int splitk, int packed, int nw, int na, int xcd_rows, int chain, int reps, double* us_out, int act) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  void* out_t = nullptr; float* out32 = nullptr; float* bias = nullptr;
  std::vector<void*> W(nw), Av(na);  // na > 1: the activation operand is never L2-resident from the previous launch (in-situ behaviour)
  const int npad = (N + 63) / 64 * 64;
  int rc = 0;
  for (int i = 0; i < na && !rc; ++i) rc = dev_bf16(ar, &Av[i], (size_t)(M + 8) * (K / taps), 1u + i);
  if (!rc) rc = ar.alloc(&out_t, (size_t)M * N * 2);
  if (!rc) rc = ar.alloc_t(&out32, (size_t)std::max(splitk, 1) * M * N);
  if (!rc) rc = dev_f32(ar, &bias, N, 5u);
  for (int i = 0; i < nw && !rc; ++i) rc = dev_bf16(ar, &W[i], (size_t)npad * K, 77u + i);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      GemmArgs g = gemm_args(Av[i % na], K / taps, W[i % nw], K, M, N, K);
      g.taps = taps; g.seq_len = seq_len > 0 ? seq_len : M; g.splitk = splitk; (void)packed; g.xcd_rows = xcd_rows;
      if (splitk > 1) { g.out_f32 = out32; g.ldo32 = N; }
      else { g.bias = bias; g.out_t = out_t; g.ldot = N; g.act = act; }
      TT_TRY(gemm_launch(DT_BF16, EPI_STD, g, s));
    }
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

}  // extern "C"


__device__ __forceinline__ float kb_dot8_fwd(Vec<bf16>::x8 a, Vec<bf16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}
// --------------------------------------------------------------------------------------------------------------------
// Round 6: GEMV-shaped decode GEMM for M <= 8 rows (the streaming path decodes ONE sequence; DESIGN 5.13: the regime where the row tile of an MFMA
// kernel is mostly padding).  A workgroup of 4 waves owns 16 output columns; a wave owns 4 of them and streams their W rows once (K / 512 sixteen-byte
// loads per lane per row, ALL requested before the first use), the M activation rows sit in registers, products on v_dot2, one cross-lane sum per
// (row, column).  No LDS, no barrier, W bytes only (32 KB per workgroup at K = 1024 against 96 KB for the 32 x 16 MFMA tile).  out f32 [M][N] + bias.
template <int MR, int KC>  // MR rows of A, KC = K / 512 chunks per lane
__global__ __launch_bounds__(256) void gemv_probe_kernel(const bf16* A, const bf16* W, const float* bias, float* out, int M, int N, int K) {
  typedef Vec<bf16>::x8 x8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16 + wave * 4;
  x8 w[4][KC];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16* wr = W + (size_t)min(n0 + c, N - 1) * K + lane * 8;
#pragma unroll
    for (int k = 0; k < KC; ++k) w[c][k] = *(const x8*)(wr + k * 512);
  }
  x8 a[MR][KC];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int k = 0; k < KC; ++k) a[r][k] = *(const x8*)(A + (size_t)min(r, M - 1) * K + lane * 8 + k * 512);
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) acc = kb_dot8_fwd(a[r][k], w[c][k], acc);
      acc = wave_sum(acc);
      if (lane == 0 && r < M && n0 + c < N) out[(size_t)r * N + n0 + c] = acc + bias[n0 + c];
    }
}

extern "C" int tt_kb_gemv_probe(int M, int N, int K, int nw, int na, int chain, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  TT_REQUIRE((K == 1024 || K == 4096) && M >= 1 && M <= 8, "gemv probe: K 1024 / 4096, M <= 8");
  float* out = nullptr; float* bias = nullptr;
  std::vector<void*> W(nw), Av(na);
  int rc = 0;
  for (int i = 0; i < na && !rc; ++i) rc = dev_bf16(ar, &Av[i], (size_t)(M + 8) * K, 1u + i);
  if (!rc) rc = ar.alloc_t(&out, (size_t)M * N + 64);
  if (!rc) rc = dev_f32(ar, &bias, N + 64, 5u);
  for (int i = 0; i < nw && !rc; ++i) rc = dev_bf16(ar, &W[i], (size_t)(N + 64) * K, 77u + i);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      const bf16* a_ = (const bf16*)Av[i % na];
      const bf16* w_ = (const bf16*)W[i % nw];
      const dim3 grid((N + 15) / 16);
      if (K == 1024) {
        if (M <= 1) gemv_probe_kernel<1, 2><<<grid, 256, 0, s>>>(a_, w_, bias, out, M, N, K);
        else if (M <= 4) gemv_probe_kernel<4, 2><<<grid, 256, 0, s>>>(a_, w_, bias, out, M, N, K);
        else gemv_probe_kernel<8, 2><<<grid, 256, 0, s>>>(a_, w_, bias, out, M, N, K);
      } else {
        if (M <= 1) gemv_probe_kernel<1, 8><<<grid, 256, 0, s>>>(a_, w_, bias, out, M, N, K);
        else gemv_probe_kernel<4, 8><<<grid, 256, 0, s>>>(a_, w_, bias, out, M, N, K);
      }
    }
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

// --------------------------------------------------------------------------------------------------------------------
// Decode attention experiment: same score phase as the product kernel; PV phase with 16-byte V loads (8 lanes per key
// row of 128 B, 8 keys per wave instruction = 1 KiB like the K loads) instead of 8-byte loads (512 B per instruction).
__device__ __forceinline__ float kb_dot8(Vec<bf16>::x8 a, Vec<bf16>::x8 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}

template <int VU>  // V row sets of 8 keys per register set
__global__ __launch_bounds__(256, 4) void decode_attn_v2_kernel(DecodeAttnArgs a, int ctx_cap) {
  typedef Vec<bf16>::x8 x8;
  typedef bf16 T;
  extern __shared__ __attribute__((aligned(16))) float sc_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = min((int)blockIdx.x * 4 + wave, a.B * a.heads - 1);
  const int b = pair / a.heads, h = pair % a.heads;
  const int tgen = *a.step + 1;
  const int P1 = a.P1;
  const int ctx = P1 + tgen;
  float* sc = sc_all + (size_t)wave * ctx_cap;
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
    const int nsp = (P1 + 63) >> 6, nso = (tgen + 63) >> 6;
#pragma unroll 1
    for (int sl0 = 0; sl0 < nsp + nso; sl0 += 2) {
      x8 kk[2][8];
      int key[2];
      bool live[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int slot = min(sl0 + u, nsp + nso - 1);
        const bool pre = slot < nsp;
        const int k = (pre ? slot : slot - nsp) * 64 + lane;
        const int lim = pre ? P1 : tgen;
        const int kcl = min(k, lim - 1);
        const char* base = (const char*)(pre ? kp : kc);
        const unsigned off = (pre ? (unsigned)kcl * 64u : (unsigned)kcl * 8u) * 2u;
        const unsigned cs = (pre ? 8u : (unsigned)a.tmax * 8u) * 2u;
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
        for (int c = 0; c < 8; ++c) sv = kb_dot8(qk[c], kk[u][c], sv);
        if (live[u]) { sc[key[u]] = sv; mx = fmaxf(mx, sv); }
      }
    }
  }
  // PV: lane -> (key sub-index kk = lane >> 3, channel group cg = lane & 7: 8 channels = 16 bytes)
  const int kk8 = lane >> 3, cg = lane & 7;
  constexpr int KEYS = 8 * VU;
  const int nvp = (P1 + KEYS - 1) / KEYS, nvo = (tgen + KEYS - 1) / KEYS, nit = nvp + nvo;
  auto load_v = [&](x8 (&t)[VU], int it) {
    const int itc = min(it, nit - 1);
    const bool pre = itc < nvp;
    const char* base = (const char*)(pre ? vp : vc);
    const int k0 = (pre ? itc : itc - nvp) * KEYS + kk8, lim = pre ? P1 : tgen;
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const unsigned jc = (unsigned)min(k0 + 8 * u, lim - 1);
      t[u] = *(const x8*)(base + (jc * 64u + (unsigned)cg * 8u) * 2u);
    }
  };
  x8 ta[VU], tb[VU];
  load_v(ta, 0);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < ctx; j += 64) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = 0.f;
  auto consume = [&](const x8 (&t)[VU], int it) {
    const bool pre = it < nvp;
    const int k0 = (pre ? it : it - nvp) * KEYS + kk8, lim = it < nit ? (pre ? P1 : tgen) : 0;
    const float* scs = sc + (pre ? 0 : P1);
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int j = k0 + 8 * u;
      const float pj = j < lim ? scs[j] : 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] += pj * (float)t[u][c];
    }
  };
#pragma unroll 1
  for (int it = 0; it < nit; it += 2) {
    load_v(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    consume(ta, it);
    __builtin_amdgcn_sched_barrier(0);
    load_v(ta, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    consume(tb, it + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    o[c] += __shfl_xor(o[c], 8, 64);
    o[c] += __shfl_xor(o[c], 16, 64);
    o[c] += __shfl_xor(o[c], 32, 64);
  }
  if ((int)blockIdx.x * 4 + wave < a.B * a.heads && kk8 == 0) {
    const float inv = 1.0f / sum;
    x8 r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = (T)(o[c] * inv);
    *(x8*)((T*)a.out + (size_t)b * a.heads * 64 + h * 64 + cg * 8) = r;
  }
}

extern "C" {
// B sequences x heads, tgen generated keys, prefix P1; `nl` distinct per-layer caches visited round-robin (cold KV).
// variant 0 = product kernel, 1 = v2 with 4 row sets (32 keys / iteration), 2 = v2 with 6 row sets (48 keys / iteration).
int tt_kb_decode_attn(int variant, int B, int heads, int P1, int tgen, int tmax, int nl, int chain, int reps, double* us_out, double* maxdiff) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  void* q = nullptr; void* kp = nullptr; void* vp = nullptr; void* out = nullptr; void* out_ref = nullptr; int* step = nullptr;
  std::vector<void*> kc(nl), vc(nl);
  const size_t per = (size_t)B * heads * tmax * 64;
  int rc = dev_bf16(ar, &q, (size_t)B * heads * 64, 3u);
  if (!rc) rc = dev_bf16(ar, &kp, (size_t)heads * P1 * 64 + 64, 4u);
  if (!rc) rc = dev_bf16(ar, &vp, (size_t)heads * P1 * 64 + 64, 5u);
  if (!rc) rc = ar.alloc(&out, (size_t)B * heads * 64 * 2);
  if (!rc) rc = ar.alloc(&out_ref, (size_t)B * heads * 64 * 2);
  if (!rc) rc = ar.alloc_t(&step, 4);
  for (int i = 0; i < nl && !rc; ++i) {
    rc = dev_bf16(ar, &kc[i], per + 64, 100u + i);
    if (!rc) rc = dev_bf16(ar, &vc[i], per + 64, 200u + i);
  }
  const int st = tgen - 1;
  if (!rc) TT_CHECK_HIP(hipMemcpy(step, &st, sizeof(int), hipMemcpyHostToDevice));
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  auto launch = [&](int var, int layer, void* o, hipStream_t s) -> int {
    DecodeAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.kp = kp; a.vp = vp; a.P1 = P1; a.kc = kc[layer]; a.vc = vc[layer]; a.tmax = tmax; a.step = step; a.host_tgen = tgen;
    a.out = o; a.B = B; a.heads = heads;
    if (var == 0 || var >= 10) {
      a.variant = var >= 10 ? var - 9 : 0;  // 10: per-wave prefix kernel, 11 / 12: shared-prefix kernel with 16 / 4 sequences per workgroup
      return decode_attention_launch(DT_BF16, a, s);
    }
    const int ctx_cap = P1 + tmax;
    const size_t smem = (size_t)4 * ctx_cap * sizeof(float);
    const int blocks = cdiv(B * heads, 4);
    if (var == 1) decode_attn_v2_kernel<4><<<blocks, 256, smem, s>>>(a, ctx_cap);
    else decode_attn_v2_kernel<6><<<blocks, 256, smem, s>>>(a, ctx_cap);
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  };
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) TT_TRY(launch(variant, i % nl, out, s));
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  if (!rc && maxdiff) {  // agreement with the product kernel on layer 0
    rc = launch(10, 0, out_ref, gt.s);
    if (!rc) rc = launch(variant, 0, out, gt.s);
    if (!rc && hipStreamSynchronize(gt.s) != hipSuccess) rc = -2;
    if (!rc) {
      std::vector<unsigned short> x((size_t)B * heads * 64), y(x.size());
      TT_CHECK_HIP(hipMemcpy(x.data(), out, x.size() * 2, hipMemcpyDeviceToHost));
      TT_CHECK_HIP(hipMemcpy(y.data(), out_ref, y.size() * 2, hipMemcpyDeviceToHost));
      double md = 0;
      for (size_t i = 0; i < x.size(); ++i) {
        unsigned ux = (unsigned)x[i] << 16, uy = (unsigned)y[i] << 16;
        float fx, fy;
        memcpy(&fx, &ux, 4); memcpy(&fy, &uy, 4);
        md = std::max(md, (double)fabsf(fx - fy));
      }
      *maxdiff = md;
    }
  }
  gt.destroy();
  ar.release();
  return rc;
}
}  // extern "C"


// --------------------------------------------------------------------------------------------------------------------
// Per-CU load-path probe: every workgroup streams `bytes_per_wg` from a `footprint`-byte buffer (small footprint => L2 /
// Infinity-Cache resident) through one of the load paths a GEMM operand can take.
//   mode 0: global_load_lds_dwordx4 (direct to LDS, 1 KiB per wave instruction)
//   mode 1: global_load_dwordx4 to VGPRs, full 128-B lines (8 lanes per line)
//   mode 2: global_load_dwordx4 to VGPRs, MFMA-fragment shaped (16 rows x 64 B per instruction, row stride 2 KiB)
template <int MODE, int NW, int UNR = 8>
__global__ __launch_bounds__(NW * 64) void bw_probe_kernel(const char* buf, size_t footprint, size_t bytes_per_wg, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t wg_base = ((size_t)blockIdx.x * 1315423911ull * 4096) % footprint;  // scattered start, 4 KiB aligned
  const size_t per_iter = (size_t)NW * UNR * 1024;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t done = 0; done < bytes_per_wg; done += per_iter) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        size_t off = (wg_base + done + ((size_t)(u * NW + wave)) * 1024 + lane * 16) % footprint;
        __builtin_amdgcn_global_load_lds((gbl_void_k*)(buf + off), (lds_void_k*)(smem_raw + ((u * NW + wave) % 32) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNR) : "memory");
    } else {
      f32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        size_t off;
        if (MODE == 1) off = (wg_base + done + ((size_t)(u * NW + wave)) * 1024 + lane * 16) % footprint;
        else off = (wg_base + done + ((size_t)(u * NW + wave)) * 64 + (size_t)(lane & 15) * 2048 + (lane >> 4) * 16) % footprint;
        v[u] = *(const f32x4*)(buf + off);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

extern "C" int tt_kb_bw_probe(int mode, int nw_waves, int nblocks, size_t footprint, size_t bytes_per_wg, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  void* buf = nullptr; float* sink = nullptr;
  int rc = dev_bf16(ar, &buf, footprint / 2 + 4096, 9u);
  if (!rc) rc = ar.alloc_t(&sink, 64);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
#define BWP(MODE, NW) bw_probe_kernel<MODE, NW><<<nblocks, NW * 64, 32 * 1024, s>>>((const char*)buf, footprint, bytes_per_wg, sink)
    // nw_waves >= 100: 16 loads in flight per lane instead of 8 (waves = nw_waves - 100)
#define BWP16(MODE, NW) bw_probe_kernel<MODE, NW, 16><<<nblocks, NW * 64, 32 * 1024, s>>>((const char*)buf, footprint, bytes_per_wg, sink)
    if (nw_waves == 16) { if (mode == 0) BWP(0, 16); else if (mode == 1) BWP(1, 16); else BWP(2, 16); }
    else if (nw_waves == 108) { if (mode == 0) BWP16(0, 8); else if (mode == 1) BWP16(1, 8); else BWP16(2, 8); }
    else if (nw_waves == 116) { if (mode == 0) BWP16(0, 16); else if (mode == 1) BWP16(1, 16); else BWP16(2, 16); }
    else if (mode == 0 && nw_waves == 4) BWP(0, 4); else if (mode == 0) BWP(0, 8);
    else if (mode == 1 && nw_waves == 4) BWP(1, 4); else if (mode == 1) BWP(1, 8);
    else if (nw_waves == 4) BWP(2, 4); else BWP(2, 8);
#undef BWP16
#undef BWP
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }, reps, us_out);
  gt.destroy();
  ar.release();
  return rc;
}


// --------------------------------------------------------------------------------------------------------------------
// What does the ACCESS PATTERN of the decode attention's per-sequence K / V stream cost against a contiguous stream of the same size and
// launch geometry?  One wave per (sequence, head), `wpb` waves per workgroup, 16 loads of 16 B per lane in flight (as decode_attn_lds_kernel),
// 32 KB per wave, `nwaves` waves per launch, a chain of launches over DIFFERENT regions (the 30 layers: nothing comes from the Infinity Cache).
//   mode 0: the product layout at t = 128 keys - K chunk-major [8 chunks][tmax keys][8 dims] (eight runs of 2 KB, 3.7 KB apart), then V
//           [128 keys][64 dims] (16 KB contiguous); K and V blocks of one (sequence, head) in two different arrays
//   mode 1: the same 32 KB as ONE contiguous run per wave (a [key][K 64 | V 64] interleaved cache would stream like this)
//   mode 2: K and V each one contiguous 16 KB run, in two arrays
template <int MODE>
__global__ __launch_bounds__(256, 4) void kv_pattern_kernel(const char* kbuf, const char* vbuf, size_t region, int tmax, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.y * gridDim.x * 4 + (size_t)blockIdx.x * 4 + wave;  // (sequence group, head, sequence) as in the product grid
  const size_t kblk = (size_t)8 * tmax * 16, vblk = (size_t)tmax * 128;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 v[16];
  if (MODE == 0) {
    const char* kb = kbuf + region + w * kblk;
    const char* vb = vbuf + region + w * vblk;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[2 * c] = *(const f32x4*)(kb + (size_t)c * tmax * 16 + lane * 16);
      v[2 * c + 1] = *(const f32x4*)(kb + (size_t)c * tmax * 16 + 1024 + lane * 16);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(vb + (size_t)u * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  } else {
    const char* b0 = MODE == 1 ? kbuf + region + w * (kblk + vblk) : kbuf + region + w * kblk;
    const char* b1 = MODE == 1 ? b0 + 16384 : vbuf + region + w * vblk;
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(b0 + (size_t)u * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(b1 + (size_t)u * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

extern "C" int tt_kb_kv_pattern(int mode, int B, int heads, int tmax, int chain, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  const size_t nw = (size_t)B * heads;
  const size_t kblk = (size_t)8 * tmax * 16, vblk = (size_t)tmax * 128;
  const size_t region = nw * (kblk + vblk) + 65536;  // per launch of the chain (both arrays are sized for the interleaved mode)
  void* kbuf = nullptr; void* vbuf = nullptr; float* sink = nullptr;
  int rc = dev_bf16(ar, &kbuf, region * chain / 2 + 4096, 5u);
  if (!rc) rc = dev_bf16(ar, &vbuf, region * chain / 2 + 4096, 6u);
  if (!rc) rc = ar.alloc_t(&sink, 64);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      const dim3 grid(heads, B / 4);
      const size_t off = (size_t)i * region;
      if (mode == 0) kv_pattern_kernel<0><<<grid, 256, 0, s>>>((const char*)kbuf, (const char*)vbuf, off, tmax, sink);
      else if (mode == 1) kv_pattern_kernel<1><<<grid, 256, 0, s>>>((const char*)kbuf, (const char*)vbuf, off, tmax, sink);
      else kv_pattern_kernel<2><<<grid, 256, 0, s>>>((const char*)kbuf, (const char*)vbuf, off, tmax, sink);
    }
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

// Round 6: the same probe with NL loads (= 8 NL keys) per burst, product layout, and ONE burst instead of two (K and V requests all issued before the first
// use): the T(t) line of a pure-load kernel of the decode attention's geometry - what is left of decode_attn_lds_kernel's intercept once the loads are free.
template <int NL, bool ONE_BURST>
__global__ __launch_bounds__(256, 4) void kv_pattern2_kernel(const char* kbuf, const char* vbuf, size_t region, int tmax, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.y * gridDim.x * 4 + (size_t)blockIdx.x * 4 + wave;
  const size_t kblk = (size_t)8 * tmax * 16, vblk = (size_t)tmax * 128;
  const char* kb = kbuf + region + w * kblk;
  const char* vb = vbuf + region + w * vblk;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 k[NL], v[NL];
  // K: NL loads = NL / 8 slots of 64 keys x 8 chunks (chunk-major runs of 1 KB per 64 keys); V: NL loads of 8 rows (1 KB contiguous each)
#pragma unroll
  for (int u = 0; u < NL; ++u) k[u] = *(const f32x4*)(kb + (size_t)(u & 7) * tmax * 16 + (size_t)(u >> 3) * 1024 + lane * 16);
  if (ONE_BURST) {
#pragma unroll
    for (int u = 0; u < NL; ++u) v[u] = *(const f32x4*)(vb + (size_t)u * 1024 + lane * 16);
  }
#pragma unroll
  for (int u = 0; u < NL; ++u) acc += k[u];
  if (!ONE_BURST) {
#pragma unroll
    for (int u = 0; u < NL; ++u) v[u] = *(const f32x4*)(vb + (size_t)u * 1024 + lane * 16);
  }
#pragma unroll
  for (int u = 0; u < NL; ++u) acc += v[u];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

extern "C" int tt_kb_kv_pattern2(int nl, int one_burst, int B, int heads, int tmax, int chain, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  const size_t nw = (size_t)B * heads;
  const size_t kblk = (size_t)8 * tmax * 16, vblk = (size_t)tmax * 128;
  const size_t region = nw * (kblk + vblk) + 65536;
  void* kbuf = nullptr; void* vbuf = nullptr; float* sink = nullptr;
  int rc = dev_bf16(ar, &kbuf, region * chain / 2 + 4096, 5u);
  if (!rc) rc = dev_bf16(ar, &vbuf, region * chain / 2 + 4096, 6u);
  if (!rc) rc = ar.alloc_t(&sink, 64);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      const dim3 grid(heads, B / 4);
      const size_t off = (size_t)i * region;
#define KVP2(NLV)                                                                                                                        \
      do {                                                                                                                                \
        if (one_burst) kv_pattern2_kernel<NLV, true><<<grid, 256, 0, s>>>((const char*)kbuf, (const char*)vbuf, off, tmax, sink);        \
        else kv_pattern2_kernel<NLV, false><<<grid, 256, 0, s>>>((const char*)kbuf, (const char*)vbuf, off, tmax, sink);                 \
      } while (0)
      if (nl == 2) KVP2(2); else if (nl == 4) KVP2(4); else if (nl == 8) KVP2(8); else if (nl == 12) KVP2(12); else KVP2(16);
#undef KVP2
    }
    TT_CHECK_HIP(hipGetLastError());
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

extern "C" int tt_kb_flash(int B, int H, int n, int causal, int relpos, int chain, int reps, double* us_out) {
  Arena ar;
  GraphTimer gt;
  TT_TRY(gt.init());
  const int n_pad = (n + 31) / 32 * 32;
  void* q = nullptr; void* k = nullptr; void* vt = nullptr; void* out = nullptr; float* rp = nullptr;
  int rc = dev_bf16(ar, &q, (size_t)B * H * n * 64 + 64, 1u);
  if (!rc) rc = dev_bf16(ar, &k, (size_t)B * H * n * 64 + 64, 2u);
  if (!rc) rc = dev_bf16(ar, &vt, (size_t)B * H * 64 * n_pad + 64, 3u);
  if (!rc) rc = ar.alloc(&out, (size_t)B * n * H * 64 * 2);
  if (!rc) rc = dev_f32(ar, &rp, (size_t)H * 129, 4u);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  if (!rc) rc = gt.run([&](hipStream_t s) -> int {
    for (int i = 0; i < chain; ++i) {
      FlashArgs f;
      memset(&f, 0, sizeof(f));
      f.q = q; f.k = k; f.vt = vt; f.out = out; f.ldo = H * 64; f.BH = B * H; f.heads = H; f.n = n; f.n_pad = n_pad; f.causal = causal;
      f.relpos = relpos ? rp : nullptr;
      TT_TRY(flash_attention_launch(DT_BF16, f, s));
    }
    return 0;
  }, reps, us_out);
  if (!rc) *us_out /= chain;
  gt.destroy();
  ar.release();
  return rc;
}

// --------------------------------------------------------------------------------------------------------------------
// Do two launch chains on two streams overlap on this device?  The decode step alternates an HBM-bound kernel (attention)
// with latency-bound ones (64 x 64 GEMMs at one wave per SIMD, row norms); cutting the candidates into row ranges on
// separate streams only pays if kernels of different queues actually share the chip.  Chains (each `chain` launches long):
//   G = the decode projection GEMM (M rows, N = K = 1024, 4 split-K slabs), A = decode attention over M sequences x 16 heads.
// out[0] G alone, [1] G || G on two streams (own activations, shared weights), [2] A alone, [3] A || G, [4] A || A,
// [5] G || G as ONE hipGraph with two parallel branches, [6] A || G as one graph with two branches: microseconds per
// replay of the whole chain(s).  Perfect overlap: [1] == [0], [3] == max([0], [2]); serialised: [1] == 2 x [0], [3] == [0] + [2].
extern "C" int tt_kb_concurrency(int M, int chain, int tgen, int reps, double* out) {
  Arena ar;
  hipStream_t s0 = nullptr, s1 = nullptr;
  hipEvent_t ea = nullptr, eb = nullptr, ef = nullptr, ej = nullptr;
  TT_CHECK_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  TT_CHECK_HIP(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  TT_CHECK_HIP(hipEventCreate(&ea));
  TT_CHECK_HIP(hipEventCreate(&eb));
  TT_CHECK_HIP(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  TT_CHECK_HIP(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  const int D = 1024, H = 16, P1 = 59, tmax = tgen + 8, NW = 8, NL = 4;
  void* A[2]; float* slab[2]; void* W[NW];
  void* q[2]; void* o[2]; void* kp = nullptr; void* vp = nullptr; void* kc[2][NL]; void* vc[2][NL]; int* step = nullptr;
  int rc = 0;
  for (int i = 0; i < 2 && !rc; ++i) {
    rc = dev_bf16(ar, &A[i], (size_t)(M + 8) * D, 11u + i);
    if (!rc) rc = ar.alloc_t(&slab[i], (size_t)4 * M * D);
    if (!rc) rc = dev_bf16(ar, &q[i], (size_t)M * D, 21u + i);
    if (!rc) rc = ar.alloc(&o[i], (size_t)M * D * 2);
    for (int l = 0; l < NL && !rc; ++l) {
      rc = dev_bf16(ar, &kc[i][l], (size_t)M * H * tmax * 64 + 64, 100u + 8 * i + l);
      if (!rc) rc = dev_bf16(ar, &vc[i][l], (size_t)M * H * tmax * 64 + 64, 200u + 8 * i + l);
    }
  }
  for (int i = 0; i < NW && !rc; ++i) rc = dev_bf16(ar, &W[i], (size_t)D * D, 77u + i);
  if (!rc) rc = dev_bf16(ar, &kp, (size_t)H * P1 * 64 + 64, 4u);
  if (!rc) rc = dev_bf16(ar, &vp, (size_t)H * P1 * 64 + 64, 5u);
  if (!rc) rc = ar.alloc_t(&step, 4);
  const int st = tgen - 1;
  if (!rc && hipMemcpy(step, &st, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = -2;
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = -2;
  auto chain_g = [&](int i, hipStream_t s) -> int {
    for (int c = 0; c < chain; ++c) {
      GemmArgs g = gemm_args(A[i], D, W[c % NW], D, M, D, D);
      g.splitk = 4; g.out_f32 = slab[i]; g.ldo32 = D;
      TT_TRY(gemm_launch(DT_BF16, EPI_STD, g, s));
    }
    return 0;
  };
  auto chain_a = [&](int i, hipStream_t s) -> int {
    for (int c = 0; c < chain; ++c) {
      DecodeAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = q[i]; a.kp = kp; a.vp = vp; a.P1 = P1; a.kc = kc[i][c % NL]; a.vc = vc[i][c % NL]; a.tmax = tmax; a.step = step; a.host_tgen = tgen;
      a.out = o[i]; a.B = M; a.heads = H;
      TT_TRY(decode_attention_launch(DT_BF16, a, s));
    }
    return 0;
  };
  auto capture = [&](hipStream_t s, auto&& fn, hipGraphExec_t* ex) -> int {
    hipGraph_t g = nullptr;
    TT_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int r = fn();
    hipError_t ce = hipStreamEndCapture(s, &g);
    if (r) return r;
    TT_CHECK_HIP(ce);
    TT_CHECK_HIP(hipGraphInstantiate(ex, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    return 0;
  };
  // x: graph for stream 0 (or null), y: graph for stream 1 (or null): `reps` replays of each, concurrently; wall time on s0's clock
  auto time_pair = [&](hipGraphExec_t x, hipGraphExec_t y, double* us) -> int {
    for (int warm = 0; warm < 2; ++warm) {
      TT_CHECK_HIP(hipEventRecord(ea, s0));
      TT_CHECK_HIP(hipEventRecord(ef, s0));
      TT_CHECK_HIP(hipStreamWaitEvent(s1, ef, 0));
      const int n = warm ? reps : 2;
      for (int r = 0; r < n; ++r) {
        if (x) TT_CHECK_HIP(hipGraphLaunch(x, s0));
        if (y) TT_CHECK_HIP(hipGraphLaunch(y, s1));
      }
      TT_CHECK_HIP(hipEventRecord(ej, s1));
      TT_CHECK_HIP(hipStreamWaitEvent(s0, ej, 0));
      TT_CHECK_HIP(hipEventRecord(eb, s0));
      TT_CHECK_HIP(hipStreamSynchronize(s0));
      TT_CHECK_HIP(hipStreamSynchronize(s1));
    }
    float ms = 0.f;
    TT_CHECK_HIP(hipEventElapsedTime(&ms, ea, eb));
    *us = 1e3 * ms / reps;
    return 0;
  };
  hipGraphExec_t g0 = nullptr, g1 = nullptr, a0 = nullptr, a1 = nullptr, gg = nullptr, ag = nullptr;
  if (!rc) rc = capture(s0, [&]() { return chain_g(0, s0); }, &g0);
  if (!rc) rc = capture(s1, [&]() { return chain_g(1, s1); }, &g1);
  if (!rc) rc = capture(s0, [&]() { return chain_a(0, s0); }, &a0);
  if (!rc) rc = capture(s1, [&]() { return chain_a(1, s1); }, &a1);
  auto forked = [&](auto&& f0, auto&& f1) -> int {  // one graph, two parallel branches
    TT_CHECK_HIP(hipEventRecord(ef, s0));
    TT_CHECK_HIP(hipStreamWaitEvent(s1, ef, 0));
    TT_TRY(f0());
    TT_TRY(f1());
    TT_CHECK_HIP(hipEventRecord(ej, s1));
    TT_CHECK_HIP(hipStreamWaitEvent(s0, ej, 0));
    return 0;
  };
  if (!rc) rc = capture(s0, [&]() { return forked([&]() { return chain_g(0, s0); }, [&]() { return chain_g(1, s1); }); }, &gg);
  if (!rc) rc = capture(s0, [&]() { return forked([&]() { return chain_a(0, s0); }, [&]() { return chain_g(1, s1); }); }, &ag);
  if (!rc) rc = time_pair(g0, nullptr, &out[0]);
  if (!rc) rc = time_pair(g0, g1, &out[1]);
  if (!rc) rc = time_pair(a0, nullptr, &out[2]);
  if (!rc) rc = time_pair(a0, g1, &out[3]);
  if (!rc) rc = time_pair(a0, a1, &out[4]);
  if (!rc) rc = time_pair(gg, nullptr, &out[5]);
  if (!rc) rc = time_pair(ag, nullptr, &out[6]);
  for (hipGraphExec_t ex : {g0, g1, a0, a1, gg, ag})
    if (ex) (void)hipGraphExecDestroy(ex);
  (void)hipEventDestroy(ea); (void)hipEventDestroy(eb); (void)hipEventDestroy(ef); (void)hipEventDestroy(ej);
  (void)hipStreamDestroy(s0); (void)hipStreamDestroy(s1);
  ar.release();
  return rc;
}
