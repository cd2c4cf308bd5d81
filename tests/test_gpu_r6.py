"""-m gpu, round 6: the decode step pinned at the CONTEXTS the benchmark runs it at, and BASELINE config #2 at its own shapes.

  * `tt_op_decode_attention` (the dominant kernel of the headline, csrc/attention.hip decode_attn_lds_kernel / decode_attn_kernel) at 16 heads x
    256 sequences, 59 shared-prefix keys, 1 .. 500 own keys against torch fp32 from the same rounded operands - every other attention form
    already had such a test (tests/test_gpu_ops.py::test_flash_attention).
  * the autoregressive engine teacher-forced for 500 steps at B = 96 ('fast' preset, ragged on the 64-row decode tiles) and B = 256 against
    the reference's own GPT2InferenceModel (tests/golden/full_ar_long.npz from oracle/make_golden_full.py::full_ar_long;
    tortoise/models/autoregressive.py:108-186): logits after 1, 63, 64, 65, 127, 128, 199, 200, 320, 499, 500 fed tokens, and the warped
    sampling distribution (HF processors, stream_generator.py:916-1000) there: total variation, nucleus-set agreement, top-1.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from oracle import make_golden_full as GF
from oracle import tortoise_oracle as O
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd.config import ARConfig
from tests.gpu_util import DTYPES, report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = ("f32", E.TT_F32, torch.float32, 2e-5)


@pytest.fixture(scope="module")
def lib():
    return E.init()


@pytest.fixture(scope="module")
def sds():
    import bench
    return bench.synthetic_weights()


def _decode_attention_reference(q, kp, vp, k_own, v_own):
    """q [B,H,64] (pre-scaled), kp / vp [H,P1,64], k_own / v_own [B,H,t,64], fp32: softmax over [prefix | own keys] (GPT2Attention._attn)."""
    B = q.shape[0]
    k = torch.cat([kp[None].expand(B, -1, -1, -1), k_own], dim=2)
    v = torch.cat([vp[None].expand(B, -1, -1, -1), v_own], dim=2)
    w = torch.einsum("bhd,bhkd->bhk", q, k)
    return torch.einsum("bhk,bhkd->bhd", torch.softmax(w, dim=-1), v)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES + [F32])
@pytest.mark.parametrize("B,P1,tgen,tmax", [(256, 59, 1, 208), (256, 59, 64, 208), (256, 59, 65, 208), (256, 59, 200, 208), (256, 59, 500, 508),
                                            (96, 59, 127, 208), (32, 59, 128, 208), (7, 33, 3, 16)])
def test_decode_attention_operator(lib, name, dt, tdt, tol, B, P1, tgen, tmax):
    """Own keys 1 / 64 / 65 / 127 / 128 straddle the 64-key slots of the score loop and the two-slot prefetch; 200 = the benchmark's last
    step, 500 = the maximum decode length; B = 96 / 32 / 7: ragged and small batches (surplus waves of the last workgroup).  Variants:
    the shape's default (LDS-staged shared prefix, 4 sequences per workgroup), 16 per workgroup, and the per-wave prefix kernel."""
    H = 16
    g = torch.Generator().manual_seed(B * 1000 + tgen)
    q = (torch.randn(B, H, 64, generator=g) * 0.125 * 2).to(tdt)
    kp = (torch.randn(H, P1, 64, generator=g) * 2).to(tdt)
    vp = torch.randn(H, P1, 64, generator=g).to(tdt)
    k_own = (torch.randn(B, H, tgen, 64, generator=g) * 2).to(tdt)
    v_own = torch.randn(B, H, tgen, 64, generator=g).to(tdt)
    # cache layouts (include/tortoise_mi355x.h): keys [B][H][8 chunks][tmax][8], values [B][H][tmax][64]; slots >= tgen hold garbage on purpose
    kc = torch.full((B, H, 8, tmax, 8), 1e4).to(tdt)
    kc[:, :, :, :tgen] = k_own.reshape(B, H, tgen, 8, 8).permute(0, 1, 3, 2, 4)
    vc = torch.full((B, H, tmax, 64), -1e4).to(tdt)
    vc[:, :, :tgen] = v_own
    want = _decode_attention_reference(q.float(), kp.float(), vp.float(), k_own.float(), v_own.float()).reshape(B, H * 64)
    dq, dkp, dvp, dkc, dvc = (t.cuda().contiguous() for t in (q.reshape(B, H * 64), kp, vp, kc, vc))
    for variant in ((0, 2, 1) if dt != E.TT_F32 else (0,)):
        out = torch.zeros(B, H * 64, device="cuda", dtype=tdt)
        E.check(lib.tt_op_decode_attention(dt, E.ptr(dq), E.ptr(dkp), E.ptr(dvp), P1, E.ptr(dkc), E.ptr(dvc), tmax, tgen, E.ptr(out), B, H, variant, None))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        # the output is rounded to T once (f32 accumulation): 2^-9 (bf16) / 2^-12 (fp16) relative per element
        report(f"decode attention {name} B={B} P1={P1} own keys={tgen} variant={variant}", out.float(), want,
               {"bf16": 4e-3, "f16": 6e-4, "f32": 2e-6}[name])


def _warped(logits, ids):
    return torch.softmax(O.warp_logits(logits.float(), ids, 2.0, 0.8, 50, 0.8), dim=-1)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES + [F32])
@torch.no_grad()
def test_full_width_decode_at_the_benchmarked_contexts(sds, name, dt, tdt, tol):
    """8 distinct rows, teacher-forced for 500 steps, placed cyclically into B = 96 and B = 256 (fp32 verification mode: B = 96, 200 steps)."""
    g = np.load(os.path.join(GOLD, "full_ar_long.npz"))
    cfg = ARConfig()
    text, auto, _ = GF.prompt()
    toks = GF.arl_tokens()
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False  # suppressed (-1e9) in the benchmark weights
    f32 = dt == E.TT_F32
    steps = 200 if f32 else GF.ARL_STEPS
    st = stages.ArStage(sds["autoregressive"], cfg, dtype=dt, max_batch=96 if f32 else 256, max_text=80, max_new_tokens=steps + 8, max_latent_candidates=1)
    for B in ((96,) if f32 else (96, 256)):
        rep = B // GF.ARL_B
        st.prefill(auto, text)
        st.begin(B)
        tv_all, jac_all, top1_all, worst_rel = [], [], [], 0.0
        for s in range(steps):
            st.decode_step(toks[s].repeat(rep))
            n = s + 1
            if n not in GF.ARL_CHECK:
                continue
            want8 = torch.from_numpy(g["logits_%d" % n])
            rows = want8.shape[0]
            got = st.logits(B).cpu()
            # every copy of a row computes the same bits, wherever it sits in the batch (row tiles, split-K and attention workgroups differ)
            assert torch.equal(got[:GF.ARL_B], got[B - GF.ARL_B:]), f"a row's logits depend on its position in the decode batch (B={B}, step {n})"
            got8 = got[:rows]
            worst_rel = max(worst_rel, report(f"FULL AR logits after {n} fed tokens {name} B={B} vs reference golden", got8[:, keep], want8[:, keep],
                                              2e-5 if f32 else tol * 1.6))
            ids = torch.cat([torch.full((rows, 1), 1, dtype=torch.long), torch.full((rows, 1), cfg.start_mel_token, dtype=torch.long), toks[:n, :rows].t()], dim=1)
            p, q = _warped(got8, ids), _warped(want8, ids)
            tv_all.append(0.5 * (p - q).abs().sum(-1))
            sp, sq = p > 0, q > 0
            jac_all.append((sp & sq).sum(-1).float() / (sp | sq).sum(-1).float())
            top1_all.append((p.argmax(-1) == q.argmax(-1)).float())
        tv, jac, top1 = torch.cat(tv_all), torch.cat(jac_all), torch.cat(top1_all)
        print(f"[parity] FULL AR warped distribution at contexts {[c for c in GF.ARL_CHECK if c <= steps]} {name} B={B}: total variation mean {float(tv.mean()):.4f} "
              f"max {float(tv.max()):.4f} | nucleus-set Jaccard mean {float(jac.mean()):.4f} min {float(jac.min()):.4f} | top-1 agreement {float(top1.mean()):.4f} "
              f"| worst logits rel-L2 {worst_rel:.3e} ({len(tv)} (row, step) samples)")
        tv_mean_bound, tv_max_bound, jac_bound = {"bf16": (0.05, 0.35, 0.85), "f16": (0.01, 0.10, 0.95), "f32": (1e-4, 1e-3, 0.999)}[name]
        assert float(tv.mean()) < tv_mean_bound and float(tv.max()) < tv_max_bound and float(jac.mean()) > jac_bound
    st.close()


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M", [1, 7, 16, 32, 33, 64])
def test_skinny_decode_tiles_are_bit_identical_to_the_64x64_tile(lib, name, dt, tdt, tol, M):
    """Round 6: weight-streaming GEMMs of a small decode batch (M <= 64 rows, N >= 1024) run on 32 x 16 / 64 x 16 tiles - 192 .. 513 workgroups
    instead of 48 .. 64 (csrc/gemm_impl.h Tile).  Same MFMA, same k order per output element, same split-K ranges: every output form the decode
    step uses (f32 + bias: lm_head at the padded vocabulary 8196; split-K slabs: the projections; bias + tanh-GELU + T: c_fc; bias + residual)
    must agree BIT FOR BIT with the 64 x 64 tile (ttx_kernel_variant(TTX_GEMM_SKINNY, 0)) - which is what keeps a candidate's sampled codes
    independent of the per-rank batch it is decoded in - and both must agree with torch."""
    g = torch.Generator().manual_seed(M)
    cases = [(8196, 1024, 1, E.ACT_NONE, "f32"), (1024, 4096, 4, E.ACT_NONE, "slab"), (1024, 1024, 4, E.ACT_NONE, "slab"), (4096, 1024, 1, E.ACT_GELU_TANH, "t"),
             (1024, 1024, 1, E.ACT_NONE, "res")]
    for (N, K, sk, act, form) in cases:
        A = (torch.randn(M, K, generator=g)).to(tdt).cuda()
        Wt = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt).cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        outs = []
        for skinny in (1, 0):
            prev = lib.ttx_kernel_variant(E.TTX_GEMM_SKINNY, skinny)
            try:
                o32 = torch.zeros(max(sk, 1), M, N, device="cuda") if form != "t" else None
                ot = torch.zeros(M, N, device="cuda", dtype=tdt) if form == "t" else None
                E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(Wt), K, M, N, K, 1, 0, sk, E.ptr(bias) if form != "slab" else None, act,
                                       E.ptr(res) if form == "res" else None, E.ptr(o32), E.ptr(ot), None))
                torch.cuda.synchronize()
            finally:
                lib.ttx_kernel_variant(E.TTX_GEMM_SKINNY, prev)
            outs.append(o32 if o32 is not None else ot)
        assert torch.equal(outs[0], outs[1]), f"skinny tile differs from the 64 x 64 tile: M={M} N={N} K={K} splitk={sk} {form}"
        ref = A.float() @ Wt.float().t()
        if form == "slab":
            got = outs[0].sum(0)
        elif form == "t":
            got, ref = outs[0].float(), torch.nn.functional.gelu(ref + bias, approximate="tanh")
        elif form == "res":
            got, ref = outs[0][0], ref + bias + res
        else:
            got, ref = outs[0][0], ref + bias
        report(f"skinny GEMM {name} M={M} N={N} K={K} splitk={sk} {form}", got, ref, 2e-5 if form != "t" else {"bf16": 4e-3, "f16": 6e-4}[name])


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@torch.no_grad()
def test_grouped_clvp_scores_equal_scoring_every_utterance_alone(sds, name, dt, tdt, tol):
    """tt_clvp_score_groups (long-form reading: the utterances of a tts_many wave ranked in ONE speech-tower pass; tortoise/read.py:66-71 +
    api.py:460-477 score chunk after chunk): 3 utterances with texts of different lengths x 8 candidates of 40 codes on the 768 / 12 / 20
    towers - every score is bit-identical to tt_clvp_score on that utterance alone, and both track the oracle."""
    from tortoise_tts_amd.config import CLVPConfig
    cfg = CLVPConfig()
    g = torch.Generator().manual_seed(12)
    texts = [torch.randint(1, 255, (1, T), generator=g) for T in (23, 61, 40)]
    N, n = 8, 40
    codes = torch.randint(0, 8192, (3 * N, n), generator=g)
    st = stages.ClvpStage(sds["clvp"], cfg, dtype=dt, max_rows=3 * N * n)
    grouped = st.score_groups(texts, codes).cpu()
    alone = torch.cat([st.score(t, codes[i * N:(i + 1) * N]).cpu() for i, t in enumerate(texts)])
    small = stages.ClvpStage(sds["clvp"], cfg, dtype=dt, max_rows=N * n)  # capacity of ONE utterance: the grouped call falls back to per-utterance passes
    assert torch.equal(small.score_groups(texts, codes).cpu(), alone)
    small.close()
    st.close()
    assert torch.equal(grouped, alone), "a candidate's CLVP score depends on which utterances share its speech-tower pass"
    from tests.gpu_util import quantize_sd
    sdq = quantize_sd(sds["clvp"], tdt)
    want = torch.cat([O.clvp_score(sdq, cfg, t.long().repeat(N, 1), codes[i * N:(i + 1) * N]) for i, t in enumerate(texts)])
    err = float((grouped - want).abs().max()) / float(want.abs().max())
    print(f"[parity] grouped CLVP scores (3 utterances x {N} candidates) {name} vs oracle: max_abs/scale={err:.3e} (tol {tol:.1e})")
    assert err < tol


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_gemv_decode_gemm_operator(lib, name, dt, tdt, tol, M):
    """csrc/gemv.hip (round 6): the decode step's GEMMs for handles of <= 4 sequences (the streaming engine decodes ONE; reference work:
    tortoise/models/autoregressive.py:150-163 per token, api_fast.py:389-420) - W rows streamed once per workgroup, the rows in registers,
    v_dot2 products, one cross-lane sum per output.  Every epilogue against torch fp32 from the same rounded operands: f32 + bias at the
    padded vocabulary (lm_head), in-place residual update (the projections, K = 1024 and 4096), bias + tanh-GELU + T (c_fc)."""
    g = torch.Generator().manual_seed(40 + M)
    for (N, K, epi) in ((8196, 1024, 0), (1024, 1024, 1), (1024, 4096, 1), (4096, 1024, 2)):
        A = torch.randn(M, K, generator=g).to(tdt).cuda()
        Wt = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt).cuda()
        bias = torch.randn(N, generator=g).cuda()
        x0 = torch.randn(M, N, generator=g).cuda()
        o32 = x0.clone()
        ot = torch.zeros(M, N, device="cuda", dtype=tdt)
        E.check(lib.tt_op_gemv(dt, E.ptr(A), E.ptr(Wt), M, N, K, E.ptr(bias), epi, E.ptr(o32), E.ptr(ot), None))
        torch.cuda.synchronize()
        ref = A.float() @ Wt.float().t() + bias
        if epi == 0:
            report(f"gemv {name} M={M} N={N} K={K} f32 + bias", o32, ref, 2e-5)
        elif epi == 1:
            report(f"gemv {name} M={M} N={N} K={K} residual update", o32, x0 + ref, 2e-5)
        else:
            report(f"gemv {name} M={M} N={N} K={K} bias + gelu -> T", ot.float(), torch.nn.functional.gelu(ref, approximate="tanh"), {"bf16": 4e-3, "f16": 6e-4}[name])
    # the same kernel with the LayerNorm in front of c_fc inside it (f32 residual rows in; the normalised row is rounded to T as the row-norm kernel's output is)
    x = (torch.randn(M, 1024, generator=g) * 3 + 0.5).cuda()
    gam = (1 + 0.2 * torch.randn(1024, generator=g)).cuda()
    bet = (0.1 * torch.randn(1024, generator=g)).cuda()
    Wt = (torch.randn(4096, 1024, generator=g) / 32).to(tdt).cuda()
    bias = torch.randn(4096, generator=g).cuda()
    ot = torch.zeros(M, 4096, device="cuda", dtype=tdt)
    E.check(lib.tt_op_gemv_ln(dt, E.ptr(x), E.ptr(gam), E.ptr(bet), 1e-5, E.ptr(Wt), M, 4096, E.ptr(bias), E.ptr(ot), None))
    torch.cuda.synchronize()
    h = torch.nn.functional.layer_norm(x, (1024,), gam, bet, 1e-5).to(tdt).float()
    report(f"gemv {name} M={M} LayerNorm inside, bias + gelu -> T", ot.float(), torch.nn.functional.gelu(h @ Wt.float().t() + bias, approximate="tanh"), {"bf16": 4e-3, "f16": 6e-4}[name])


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@torch.no_grad()
def test_streaming_handle_decodes_on_the_gemv_path_within_the_parity_bars(sds, lib, name, dt, tdt, tol):
    """A handle created for <= 4 sequences (api_fast's engine: max_batch = 1) runs the decode step GEMV-shaped.  Teacher-forced on rows of
    tests/golden/full_ar_long.npz for 130 steps (two 64-key slots of the attention kernel): logits against the reference's own
    GPT2InferenceModel at the usual 16-bit bars, against the MFMA path (ttx_kernel_variant(TTX_AR_GEMV, 0): another summation order, so operand
    noise apart), and generation on the handle is deterministic and chunked == one-shot (the handle never changes kernels between calls).
    Levels of TTX_AR_GEMV: 2 (default) = the layer norms computed inside the QKV / c_fc GEMVs, 1 = behind row-norm launches, 0 = MFMA tiles."""
    g = np.load(os.path.join(GOLD, "full_ar_long.npz"))
    cfg = ARConfig()
    text, auto, _ = GF.prompt()
    toks = GF.arl_tokens()
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False
    got = {}
    LEVEL = {2: "GEMV + inner norms", 1: "GEMV", 0: "MFMA"}
    for gemv in (2, 1, 0):
        prev = lib.ttx_kernel_variant(E.TTX_AR_GEMV, gemv)
        try:
            st = stages.ArStage(sds["autoregressive"], cfg, dtype=dt, max_batch=4, max_text=80, max_new_tokens=140, max_latent_candidates=1)
        finally:
            lib.ttx_kernel_variant(E.TTX_AR_GEMV, prev)
        for B in (1, 3):
            st.prefill(auto, text)
            st.begin(B)
            for s in range(130):
                st.decode_step(toks[s, :B])
                if s + 1 in (1, 64, 65, 128):
                    lgts = st.logits(B).cpu()
                    got[(gemv, B, s + 1)] = lgts
                    want = torch.from_numpy(g["logits_%d" % (s + 1)])[:B]
                    report(f"streaming-size handle ({LEVEL[gemv]} decode GEMMs) {name} B={B} logits after {s + 1} tokens vs reference golden",
                           lgts[:, keep], want[:, keep], tol * 1.6)
        if gemv:
            st.prefill(auto, text)
            a, n = st.generate(1, 48, seed=5)
            a = a.clone()
            st.prefill(auto, text)
            b, _ = st.generate(1, 48, seed=5)
            assert torch.equal(a, b), "generation on the GEMV path is not deterministic"
            st.prefill(auto, text)
            last = None
            for c, _fin in st.generate_stream(1, 48, 16, first_chunk=16, seed=5):
                last = c.clone()
            assert last.shape[1] == a.shape[1] and torch.equal(last, a), "chunked decoding differs from one-shot on the GEMV path"
        st.close()
    for B in (1, 3):
        for n_ in (1, 64, 65, 128):
            report(f"GEMV vs MFMA decode GEMMs {name} B={B} after {n_} tokens", got[(1, B, n_)][:, keep], got[(0, B, n_)][:, keep], tol)
            report(f"GEMV with inner norms vs GEMV behind norm launches {name} B={B} after {n_} tokens", got[(2, B, n_)][:, keep], got[(1, B, n_)][:, keep], tol)


@pytest.mark.parametrize("V,k,quant,top_p", [(8194, 50, 0, 0.8), (8194, 50, 8, 1.0), (8194, 50, 2, 1.0), (10000, 50, 0, 0.8), (10000, 200, 4, 1.0), (300, 50, 0, 0.8), (8194, 1, 0, 0.8),
                                              (8194, 256, 0, 0.8), (8194, 50, 0, 0.3)])
def test_sampler_tokens_equal_the_oracle_over_vocabularies_ties_and_k(lib, V, k, quant, top_p):
    """csrc/sampling.hip after round 6's rework (ballot search of the top-k bound, wave-aggregated candidate compaction, ONE all-pairs count shared by
    the four waves, the top-p tail of <= 64 survivors in one wave): the sampled token of 32 rows equals the oracle's restatement of the HF warpers
    (stream_generator.py:916-1000 order: repetition penalty, temperature, top-k with ties kept, top-p, multinomial as argmax(p / q)) with the same
    Exp(1) draws - at the model's vocabulary (33 entries per thread) and a larger one (40), with logits quantised to 1/quant so that ties straddle the
    k-th score (more than 64 survivors: the plateau tail), with -0.0 next to +0.0, and at k = 1 / 256.  The tie cases run without top-p: WHICH of several
    equal scores a nucleus cut removes first is the sort's tie order - unspecified in the reference (torch.sort on the device, not stable); the kernel's rule
    is (score descending, token ascending), the CPU oracle's the opposite, and 1 - 2 rows of 32 differ there with the round-5 kernel and this one alike."""
    from oracle import tortoise_oracle as O
    g = torch.Generator().manual_seed(1000 + V + 7 * k + quant)
    B = 32
    logits = torch.randn(B, V, generator=g) * 3
    if quant:
        logits = torch.round(logits * quant) / quant
        logits[logits == 0] = -0.0
        logits[:, ::2][logits[:, ::2] == 0] = 0.0
    ids = torch.randint(0, V, (B, 24), generator=g)
    q = torch.empty(1, B, V).exponential_(1, generator=g)
    scores = O.warp_logits(logits, ids, 2.0, 0.8, k, top_p)
    want = O.multinomial_from_exponential(torch.softmax(scores, -1), q[0])
    seen_np = np.zeros((B, (V + 31) // 32), dtype=np.uint32)
    for b in range(B):
        for t in ids[b].tolist():
            seen_np[b, t >> 5] |= np.uint32(1 << (t & 31))
    seen = torch.from_numpy(seen_np.view(np.int32)).cuda()
    s = E.Sampling()
    s.temperature, s.top_p, s.repetition_penalty, s.top_k, s.seed, s.row_offset = 0.8, top_p, 2.0, k, 0, 0
    qd = q.cuda()
    s.exp_noise = E.ptr(qd)
    unfinished = torch.ones(B, dtype=torch.int32, device="cuda")
    codes = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
    ld = logits.cuda()
    E.check(lib.tt_op_sample(E.ptr(ld), V, B, V, E.ptr(seen), C.byref(s), 0, E.ptr(unfinished), V - 1, E.ptr(codes), 4, None))
    got = codes[:, 0].cpu().long()
    kept = int(torch.isfinite(scores).sum(-1).max())
    print(f"[parity] sampler V={V} k={k} quant={quant} top_p={top_p}: {int((got == want).sum())} / {B} tokens equal the oracle's; up to {kept} tokens kept in a row")
    assert torch.equal(got, want)
