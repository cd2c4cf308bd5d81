// MFMA GEMM / conv1d-as-GEMM for gfx950.  out[m][n] = epilogue(sum_k A[m][k] * W[n][k]).
//
// A is token-major ([rows][channels], K contiguous); W is [N][K] with K contiguous (the layout torch
// Linear / Conv1d(k=1) already use; HF Conv1D weights are transposed once at pack time).  A conv1d
// with `taps` taps over sequences of `seq_len` rows is the same GEMM with a virtual
// K = taps * cin whose k-tile `tap` reads row s + tap - taps/2 (zero outside the sequence), so
// the k3/k5 convolutions of DiffusionTts / KernelPredictor never materialise an im2col buffer.
#pragma once
#include "common.h"

namespace tt {

enum EpiKind { EPI_STD = 0, EPI_QKV_HEADS = 1, EPI_QKV_DECODE = 2, EPI_GEGLU = 3 };

struct GemmArgs {
  // operands
  const void* A;
  int lda;  // elements between consecutive A rows
  const void* W;
  int ldw;  // elements between consecutive W rows (>= K)
  // optional second activation source (plain GEMMs only): k >= k_split reads A2[m][k - k_split] instead of A[m][k], i.e.
  // out = [A | A2] @ W^T without materialising the concatenation.  a2_slot (device int) selects a2_slot_stride-element
  // blocks of A2 at kernel start (hipGraph replay: the sampler step counter).
  const void* A2;
  int lda2, k_split;
  const int* a2_slot;
  size_t a2_slot_stride;
  int M, N, K;
  int taps;     // 1 = plain GEMM
  int dilation; // rows between conv taps (0 / 1: adjacent rows); tap t reads row s + (t - taps/2) * dilation
  int seq_len;  // rows per sequence (conv boundary / head-layout epilogues)
  int cin;      // K / taps
  int splitk;   // >1: raw partial sums go to out_f32 + z * M * ldo32 (EPI_STD only)
  int serial_k; // >1: "serial split-K" - out_f32 = (((res + bias) + P0) + P1) + ... over serial_k equal K ranges inside ONE launch,
                //     bit-identical to splitk = serial_k slabs folded by the row-norm kernel (needs bias, res, out_f32; no out_t / stats / taps)
  int xcd_rows; // row bands the tile grid is cut into for XCD ownership (1, 2, 4 or 8); 0 = chosen by gemm_launch from the operand sizes
  int xcd_band; // set by gemm_launch: row tiles per band
  int sk_quot, sk_rem;  // set by gemm_launch: k-tiles per split-K slab (quotient, remainder)
  // EPI_STD
  const float* bias;
  int act;
  float slope;
  const float* res;  // f32 residual, added after the activation
  int ldres;
  float* out_f32;
  int ldo32;
  void* out_t;  // T-typed copy of the result (operand of the next GEMM)
  int ldot;
  int act_t;    // ACT_LRELU: the T-typed copy alone gets LeakyReLU(slope_t) AFTER the residual add (input activation of the next conv)
  float slope_t;
  float* gn_part;  // optional: per-(row tile, 16-column strip) sum / sum-of-squares of the f32 output (see gemm.hip)
  int gn_seq;      // rows per sequence for those statistics
  int gn_ncol16;   // set by gemm_launch
  int gn_vperiod;  // padded batches: sequence b has gn_vlen[b % gn_vperiod] valid rows; the rest are left out of the statistics (0: all valid)
  int gn_vlen[32];
  // EPI_GEGLU (x-transformers GEGLU, xtransformers.py:429-437): W rows (and bias) INTERLEAVED in strips of 16 - [value 0..15 | gate 0..15 |
  //   value 16..31 | gate 16..31 | ...] (pack.py geglu_interleave) - so that a lane holds value and gate of the same output column in two
  //   neighbouring fragments: out_t[m][j] = (value_j + b) * gelu_erf(gate_j + b), ldot = N / 2.  The [M][N] projection is never written.
  // EPI_QKV_HEADS: n -> (part = n / dmodel, head = (n % dmodel) / 64, d = n % 64), m -> (b, s)
  int dmodel, heads;
  void* q;        // [b*heads + h][seq_len][64]
  void* k;        // same
  void* v;        // same, may be null
  void* vt;       // [b*heads + h][64][seq_pad], may be null
  int seq_pad;
  float q_scale;  // multiplies q (softmax scale folded into q)
  // EPI_QKV_DECODE: m = sequence b; K/V appended at position *step of the per-sequence cache
  const int* step;  // device int
  void* qbuf;       // [M][dmodel]
  void* kc;         // [b][h][8][tmax][8]  (16-byte dim chunks are key-major: coalesced lane-per-key reads)
  void* vc;         // [b][h][tmax][64]
  int tmax;
};

// GroupNorm-apply on the A path (gemm_gna.h): `a.A` is the F32 tensor the GroupNorm reads ([M][1024], lda in floats), normalised with the
// statistics partials `gemm_part` its producing GEMM left (tiles of part_rows rows), scale-shifted, activated and cast on the way into the
// MFMA loop - out = epilogue(act(GN(A) * (1 + scale) + shift) @ W^T) without the stand-alone apply launch and its 16-bit tensor.
struct GemmGnArgs {
  const float* gamma;
  const float* beta;
  const float* ss;        // scale / shift rows [2 * 1024]: sample b reads ss + (b / ss_div) * ss_stride; nullptr: none
  size_t ss_stride;
  int ss_div;
  const float* gemm_part;
  int part_rows;
  int S;                  // rows per sample (M = samples * S)
  float eps;
  int act;                // ACT_NONE / ACT_SILU
  int* guard;             // optional operand-overflow counter (non-finite statistics)
};
// true when gemm_gna_launch has a kernel for this problem (else: groupnorm_launch + gemm_launch)
bool gemm_gna_supported(int dtype, int epi, const GemmArgs& a, const GemmGnArgs& n);
int gemm_gna_launch(int dtype, int epi, const GemmArgs& a, const GemmGnArgs& n, hipStream_t stream);
int gemm_gna_stat_rows();  // rows per statistics tile its epilogue emits (the kernel's tile height: a build knob, TT_GNA_BM)

// dtype: DT_BF16 / DT_F16 (DT_F32: the slow fp32-operand verification kernel, gemm_f32.hip).  Returns 0 or a negative error (message via tt::last_error()).
int gemm_launch(int dtype, int epi, const GemmArgs& a, hipStream_t stream);
int gemm_init();  // sets dynamic-LDS attributes; called once per process
int gemm_stat_rows(const GemmArgs& a, int dtype = DT_BF16);  // rows per statistics tile of the kernel gemm_launch would pick for `a`

}  // namespace tt
