// MFMA GEMM for gfx950, host side (see gemm.h): validation, tile / XCD-band / split-K planning and the dispatch to the
// per-operand-type kernel instantiations (gemm_impl.h via gemm_bf16.hip / gemm_f16.hip).  The tile is chosen from the
// problem shape only (pick_tile): there are no run-time overrides in the product library; experiments live in
// csrc/kbench/ (built with `build.py --kbench`).
#include "gemm_impl.h"

namespace tt {

extern template int gemm_launch_typed<bf16>(int, const GemmArgs&, const GemmPlan&, hipStream_t);
extern template int gemm_launch_typed<f16>(int, const GemmArgs&, const GemmPlan&, hipStream_t);
extern template int gemm_init_typed<bf16>();
extern template int gemm_gna_launch_typed<bf16>(const GemmArgs&, const GemmPlan&, const GnaArgs&, hipStream_t);
extern template int gemm_gna_launch_typed<f16>(const GemmArgs&, const GemmPlan&, const GnaArgs&, hipStream_t);
extern template int gemm_init_typed<f16>();
template <> int gemm_launch_typed<float>(int, const GemmArgs&, const GemmPlan&, hipStream_t);  // gemm_f32.hip (verification mode)
template <> int gemm_init_typed<float>();

// Tile choice from the problem shape only (measured on MI355X, scripts/kbench.py):
//   >= 256 tiles of 256x256 : 256x256, 16 waves of 64x64, 2 stages of 64 KB (twice the flops per byte through the per-CU load
//                             path of the 128x128 tile: +20 - 60 % at the CLVP / pre-pass shapes, profiles/r02_kbench_large.txt)
//   >= 256 tiles of 128x128 : 128x128, 8 waves, 2 stages (2 blocks per CU)
//   fewer, M > 2048 (or the 3-tap statistics convolutions of one denoiser pass: shared-halo kernel) : 128x64, 8 waves, 4-stage ring
//   one denoiser pass, 1024 < M <= 2048, N <= 1024 (1x1 GEMMs, inp_block, final conv) : 64x64 - two workgroups per CU (profiles/r03_ab_geometry.txt)
//   decode / M <= 1024      : 64x64, 4 waves, 4-stage ring (weights stream from HBM: depth hides latency)
// A GEMM that emits GroupNorm statistics keeps the 128x64 tile for every M > 256: the statistics are grouped per wave tile,
// so the same tile at one and at two batch rows keeps the denoiser's conditioned row bit-identical whether it is evaluated
// alone (split tail) or batched with the conditioning-free row.
//   weight-streaming 1 x 1 GEMMs of a small decode batch (M <= 64, N >= 1024, aligned, no statistics / second source / serial fold):
//                             32x16 (M <= 32) / 64x16, 2 waves, 8-stage ring - 192 - 512 workgroups instead of 48 - 64 (gemm_impl.h Tile)
static bool skinny_ok(const GemmArgs& a, int epi) {
  if (a.M > 64 || a.N < 1024 || a.taps != 1 || a.A2 || a.gn_part || a.serial_k > 1 || a.act_t != ACT_NONE) return false;
  if (epi == EPI_QKV_DECODE) return true;
  if (epi != EPI_STD) return false;
  return (a.N & 3) == 0 && (!a.bias || ((size_t)a.bias & 15) == 0) && (!a.res || (((size_t)a.res & 15) == 0 && (a.ldres & 3) == 0)) &&
         (!a.out_f32 || (((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0)) && (!a.out_t || (((size_t)a.out_t & 7) == 0 && (a.ldot & 3) == 0));
}
bool g_gemm_skinny = true;  // ttx_kernel_variant(TTX_GEMM_SKINNY): 0 = the 64 x 64 tile for small decode batches as well (A/B runs)

static int pick_tile(const GemmArgs& a, int epi = EPI_STD) {
  if (g_gemm_skinny && skinny_ok(a, epi)) return a.M <= 32 ? TILE_32x16 : TILE_64x16;
  const long b256 = (long)cdiv(a.M, 256) * cdiv(a.N, 256) * a.splitk;
  if (b256 >= 256 && a.N >= 256 && a.M >= 2048 && a.serial_k <= 1) return TILE_256x256;  // (the pre-pass, CLVP's speech tower, a batched denoiser)
  const long b128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.splitk;
  if (a.M > 256 && b128 >= 256 && a.N > 64) return TILE_128x128;  // (N <= 64: half of a 128-wide tile would be padding)
  // one denoiser pass (1024 < M <= 2048 rows, N = 1024): 448 tiles of 64x64 put two workgroups on every CU (70 instead of 46 GB/s of
  // L2 -> LDS per CU) where 224 tiles of 128x64 leave one - in-situ A/B -1.1 % on the sampler iteration; same statistics rows (32), same bits
  if (a.taps == 1 && a.M > 1024 && a.M <= 2048 && a.N <= 1024) return TILE_64x64;
  if (a.taps > 1 && a.gn_part == nullptr && a.M > 1024 && a.M <= 2048 && a.N <= 1024) return TILE_64x64;  // inp_block, final conv: -0.9 % more
  if (a.M > 1024 || (a.M > 256 && a.gn_part != nullptr)) return TILE_128x64;
  return TILE_64x64;  // also one denoiser row (M = S <= 1024, the split diffusion tail): 2x the workgroups of 128x64
}

// rows per statistics tile (= the wave tile height TM of the kernel that pick_tile selects)
static int tile_stat_rows(int tile) { return (tile == TILE_128x128 || tile == TILE_256x256) ? 64 : 32; }  // 256x256: 4 x 4 waves; 128x128: 2 x 4; 128x64: 4 x 2; 64x64: 2 x 2

// ProfScope classes: (tile, epilogue, conv?) -> one class per kernel that actually runs
static int prof_class(int tile, int epi, bool conv, bool stats = false) {
  if (tile == TILE_32x16) return epi == EPI_QKV_DECODE ? PROF_GEMM_32x16_QKVDEC : PROF_GEMM_32x16_STD;
  if (tile == TILE_64x16) return epi == EPI_QKV_DECODE ? PROF_GEMM_64x16_QKVDEC : PROF_GEMM_64x16_STD;
  if (tile == TILE_64x64 && epi == EPI_STD && !conv && stats) return PROF_GEMM_64x64_STATS;
  const int base = tile == TILE_64x64 ? PROF_GEMM_64x64_STD : tile == TILE_128x64 ? PROF_GEMM_128x64_STD : tile == TILE_128x128 ? PROF_GEMM_128x128_STD : PROF_GEMM_256x256_STD;
  if (epi == EPI_STD) return base + (conv ? 1 : 0);
  if (epi == EPI_GEGLU) return base;  // (reported with the plain 1x1 class of its tile)
  return base + (epi == EPI_QKV_HEADS ? 2 : 3);
}

// Device argument core: tile grid, XCD row bands (minimise the bytes one XCD pulls over the fabric, A / bands + W * bands / 8),
// split-K ranges and the reciprocals the kernel divides by.
static void plan_core(const GemmArgs& a, int epi, GemmPlan& p, int force_tile = -1, int force_bm = 0, int force_bn = 0) {
  int tile = force_tile >= 0 ? force_tile : pick_tile(a, epi);
  const int bm = force_bm ? force_bm : tile == TILE_32x16 ? 32 : (tile == TILE_64x64 || tile == TILE_64x16) ? 64 : tile == TILE_256x256 ? 256 : 128;
  const int bn = force_bn ? force_bn : (tile == TILE_32x16 || tile == TILE_64x16) ? kSkinnyBN : tile == TILE_256x256 ? 256 : tile == TILE_128x128 ? 128 : 64;
  GemmCore& c = p.core;
  memset(&c, 0, sizeof(c));
  c.A = a.A; c.W = a.W; c.lda = a.lda; c.ldw = a.ldw; c.M = a.M; c.N = a.N;
  c.cin_tiles = a.cin / 64;
  c.taps_half = a.taps >> 1;
  c.dil = a.dilation > 0 ? a.dilation : 1;
  c.seq_len = a.seq_len;
  c.seq = make_fastdiv(a.seq_len > 0 ? a.seq_len : 1);
  const int gx = cdiv(a.M, bm), gy = cdiv(a.N, bn);
  c.gx = gx; c.gy = gy;
  const unsigned nwg = (unsigned)gx * gy;
  c.xq = nwg >> 3; c.xr = nwg & 7;
  int xr = a.xcd_rows;
  if (xr != 1 && xr != 2 && xr != 4 && xr != 8) {
    const double A = (double)a.M * a.cin, W = (double)a.N * a.K;
    double best = 0;
    xr = 1;
    for (int cnt = 1; cnt <= 8; cnt *= 2) {
      const double cost = A / cnt + W * cnt / 8.0;
      if (cnt == 1 || cost < best) { best = cost; xr = cnt; }
    }
  }
  if (xr > gx) xr = gx > 0 ? gx : 1;
  const int hb = cdiv(gx, xr);            // row tiles per band (the last band may be shorter)
  const int nbands = cdiv(gx, hb);
  c.hb = hb; c.last_band = nbands - 1;
  c.band = make_fastdiv((unsigned)hb * gy);
  c.hfull = make_fastdiv(hb);
  c.hlast = make_fastdiv(gx - (nbands - 1) * hb);
  const int nk_total = a.K / 64;
  c.sk_quot = nk_total / a.splitk;
  c.sk_rem = nk_total % a.splitk;
  c.A2 = a.A2; c.a2_slot = a.a2_slot; c.a2_slot_stride = a.a2_slot_stride; c.lda2 = a.lda2; c.a2_tile = a.A2 ? a.k_split / 64 : 0x7fffffff;
  p.tile = tile;
  p.splitk = a.splitk;
  // shared-halo 3-tap kernel: the denoiser's ResBlock convolutions (statistics epilogue => 128x64 tile at every M > 256, so the
  // conditioned row keeps one accumulation order whether it is evaluated alone or batched)
  const bool al16 = (a.N & 3) == 0 && (!a.bias || ((size_t)a.bias & 15) == 0) && (!a.res || (((size_t)a.res & 15) == 0 && (a.ldres & 3) == 0)) &&
                    a.out_f32 && ((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0;
  p.conv3s = epi == EPI_STD && tile == TILE_128x64 && a.taps == 3 && a.dilation <= 1 && a.splitk == 1 && a.gn_part != nullptr && a.bias != nullptr &&
             a.out_t == nullptr && a.act == ACT_NONE && a.A2 == nullptr && al16 && a.cin >= 256;
  p.prof_id = prof_class(tile, epi, a.taps > 1, a.gn_part != nullptr);
  // algorithmic work of this launch: 2*M*N*K flops; operands read once + ONE result written once (the extra split-K slabs
  // a launch writes are an implementation cost: they show up in the PMC traffic, not here)
  const bool std_epi = epi == EPI_STD;
  const double out_bytes = (double)a.M * (epi == EPI_GEGLU ? a.N / 2 : a.N) * ((std_epi && a.out_f32 ? 4.0 : 0.0) + (a.out_t || !std_epi ? 2.0 : 0.0));
  p.flops = 2.0 * a.M * a.N * a.K;
  p.bytes = ((double)a.N * a.K + (double)a.M * a.cin) * 2.0 + out_bytes + (a.res ? 4.0 * a.M * a.N : 0.0);
}

static void normalise(GemmArgs& a) {
  if (a.taps < 1) a.taps = 1;
  if (a.splitk < 1) a.splitk = 1;
  a.cin = a.K / a.taps;
}

int gemm_stat_rows(const GemmArgs& a0, int dtype) {
  GemmArgs a = a0;
  normalise(a);
  return tile_stat_rows(dtype == DT_F32 ? TILE_64x64 : pick_tile(a));  // (the fp32 verification GEMM has one tile)
}

int gemm_launch(int dtype, int epi, const GemmArgs& a0, hipStream_t stream) {
  GemmArgs a = a0;
  normalise(a);
  TT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  TT_REQUIRE(a.K % 64 == 0 && a.cin % 64 == 0, "gemm: K=%d (taps=%d) must be a multiple of 64 per tap", a.K, a.taps);
  TT_REQUIRE((a.lda % 8 == 0 && a.ldw % 8 == 0) || (dtype == DT_F32 && a.lda % 4 == 0 && a.ldw % 4 == 0), "gemm: lda=%d / ldw=%d must be multiples of 8 elements", a.lda, a.ldw);
  TT_REQUIRE(a.splitk == 1 || (epi == EPI_STD && a.out_f32 != nullptr), "gemm: split-K needs EPI_STD with an f32 slab output");
  TT_REQUIRE(a.splitk <= a.K / 64, "gemm: splitk=%d exceeds the %d k-tiles", a.splitk, a.K / 64);
  if (a.serial_k > 1)
    TT_REQUIRE(epi == EPI_STD && a.splitk == 1 && a.bias && a.res && a.out_f32 && !a.out_t && !a.gn_part && a.taps == 1 && !a.A2 && a.act == ACT_NONE &&
                   (a.K / 64) % a.serial_k == 0 && (a.N & 3) == 0 && ((size_t)a.bias & 15) == 0 && ((size_t)a.res & 15) == 0 && (a.ldres & 3) == 0 &&
                   ((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0,
               "gemm: serial split-K needs the plain aligned bias + skip + f32 form and K / 64 = %d divisible by serial_k = %d", a.K / 64, a.serial_k);
  if (a.A2) TT_REQUIRE(a.taps == 1 && a.k_split > 0 && a.k_split < a.K && a.k_split % 64 == 0 && a.lda2 % 8 == 0, "gemm: bad second activation source (k_split=%d lda2=%d)", a.k_split, a.lda2);
  TT_REQUIRE(a.act_t == ACT_NONE || (a.act_t == ACT_LRELU && epi == EPI_STD && a.out_t && a.splitk == 1), "gemm: act_t supports LeakyReLU on the T-typed output of the standard epilogue only");
  if (a.gn_part) {
    TT_REQUIRE(epi == EPI_STD && a.splitk == 1 && a.out_f32 && a.gn_seq > 0 && a.N % 16 == 0, "gemm: GroupNorm statistics need the standard epilogue, an f32 output, no split-K and N %% 16 == 0");
    a.gn_ncol16 = a.N / 16;
  }
  if (epi == EPI_GEGLU) {
    TT_REQUIRE(a.taps == 1 && a.splitk == 1 && a.out_t && a.N % 32 == 0 && a.ldot >= a.N / 2 && (a.ldot & 3) == 0 && ((size_t)a.out_t & 15) == 0 &&
                   (a.bias == nullptr || ((size_t)a.bias & 15) == 0) && !a.A2 && !a.gn_part && !a.res && !a.out_f32,
               "gemm: the GEGLU epilogue takes a plain GEMM with N %% 32 == 0 (value / gate strips interleaved), an aligned T output of N / 2 columns and nothing else");
  } else if (epi != EPI_STD) {
    TT_REQUIRE(epi == EPI_QKV_HEADS || epi == EPI_QKV_DECODE, "gemm: unknown epilogue %d", epi);
    TT_REQUIRE(a.taps == 1, "gemm: conv taps are only supported with the standard epilogue");
    TT_REQUIRE(a.dmodel % 64 == 0 && a.N == 3 * a.dmodel && a.heads * 64 == a.dmodel, "gemm: qkv epilogue needs N == 3*dmodel, head_dim 64");
    TT_REQUIRE(a.bias == nullptr || ((size_t)a.bias & 15) == 0, "gemm: qkv bias must be 16-byte aligned");
  }
  if (a.taps > 1 || epi == EPI_QKV_HEADS) TT_REQUIRE(a.seq_len > 0 && a.M % a.seq_len == 0, "gemm: M=%d is not a whole number of sequences of %d", a.M, a.seq_len);
  GemmPlan plan;
  if (dtype == DT_F32) {
    plan_core(a, epi, plan, TILE_64x64);
    plan.conv3s = false;
    return gemm_launch_typed<float>(epi, a, plan, stream);
  }
  plan_core(a, epi, plan);
  if (dtype == DT_BF16) return gemm_launch_typed<bf16>(epi, a, plan, stream);
  if (dtype == DT_F16) return gemm_launch_typed<f16>(epi, a, plan, stream);
  set_error("gemm: unknown dtype %d", dtype);
  return -1;
}

bool gemm_gna_supported(int dtype, int epi, const GemmArgs& a0, const GemmGnArgs& n) {
  GemmArgs a = a0;
  normalise(a);
  const bool common = (dtype == DT_BF16 || dtype == DT_F16) && ((size_t)a.A & 15) == 0 && (a.lda & 3) == 0 && ((size_t)n.gamma & 15) == 0 && ((size_t)n.beta & 15) == 0 &&
                      (a.ldw & 7) == 0 && a.taps == 1 && a.splitk == 1 && a.serial_k <= 1 && a.K == kGnaC && !a.A2 && a.N % kGnaBN == 0 && a.M > 256 && a.M <= 4096 &&
                      n.S >= kGnaBM && a.M % n.S == 0 && n.gemm_part && n.part_rows > 0 && (n.part_rows & (n.part_rows - 1)) == 0 && n.S >= n.part_rows && a.gn_vperiod == 0 && !n.ss;
  const bool al16 = a.bias && ((size_t)a.bias & 15) == 0 && a.out_f32 && ((size_t)a.out_f32 & 15) == 0 && (a.ldo32 & 3) == 0;
  return common && epi == EPI_STD && al16 && a.gn_part != nullptr && !a.res && !a.out_t && a.act == ACT_NONE && n.act == ACT_SILU;
}

bool g_gemm_p8 = true;  // tt_gemm_variant

int gemm_gna_launch(int dtype, int epi, const GemmArgs& a0, const GemmGnArgs& n, hipStream_t stream) {
  GemmArgs a = a0;
  normalise(a);
  TT_REQUIRE(gemm_gna_supported(dtype, epi, a0, n), "gemm_gna: unsupported problem (M=%d N=%d K=%d S=%d)", a.M, a.N, a.K, n.S);
  a.gn_ncol16 = a.N / 16;
  a.seq_len = n.S;
  GemmPlan plan;
  plan_core(a, epi, plan, TILE_64x64, kGnaBM, kGnaBN);
  plan.conv3s = false;
  plan.prof_id = PROF_GEMM_GNA;
  plan.bytes += (double)a.M * a.cin * 2.0;  // the activation rows are f32 here
  GnaArgs d;
  memset(&d, 0, sizeof(d));
  d.gamma = n.gamma; d.beta = n.beta; d.ss = n.ss; d.ss_stride = n.ss_stride; d.ss_div = n.ss_div; d.gemm_part = n.gemm_part;
  d.part_shift = 31 - __builtin_clz((unsigned)n.part_rows);
  d.S = n.S; d.eps = n.eps; d.act = n.act; d.guard = n.guard;
  d.inv_count = 1.0 / ((double)n.S * (double)(kGnaC / 32));
  if (dtype == DT_BF16) return gemm_gna_launch_typed<bf16>(a, plan, d, stream);
  return gemm_gna_launch_typed<f16>(a, plan, d, stream);
}

int gemm_gna_stat_rows() { return kGnaBM; }

int gemm_init() {
  TT_TRY(gemm_init_typed<bf16>());
  TT_TRY(gemm_init_typed<f16>());
  return 0;
}

}  // namespace tt
