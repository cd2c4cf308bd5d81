"""-m gpu: the fp32-operand VERIFICATION mode (TT_F32) against the reference's own fp32 modules at the fp32 tolerances SURVEY.md 8(c)
states - AR logits <= 1e-3 abs, denoiser eps <= 1e-4 rel-L2, waveform <= 1e-3 abs, and the mel after the real sampling schedules -
on the BENCHMARKED shapes (tests/golden/full_*.npz: 30 x 1024 GPT trunk, 55 text tokens, S = 870 denoiser positions,
768 / 12 / 20 CLVP, 870 vocoder frames).  The bf16 / fp16 tests bound the engines by operand noise (rel-L2 of a few 1e-3 .. 1e-2),
under which a logic error of the same size could hide; here every GEMM / attention operand and the KV caches stay fp32
(csrc/gemm_f32.hip, attention_f32.hip), so what is left is summation order and a handful of fast-math intrinsics: five orders of
magnitude below the operand-noise bounds.  The same host code, engines, epilogues, norms, sampler and graphs as the product run.

The one bar taken from a measurement instead of SURVEY's estimate is the final mel of a sampling schedule: SURVEY guessed <= 1e-3 abs
(mel units); the learned-range variance exponent (log beta - clipped posterior log variance, a factor of ~20 in the last steps)
amplifies last-bit differences of the model output in isolated elements, so that two fp32 CPU implementations - the reference's
p_sample_loop and the oracle - already differ by up to 5e-4 in x units (tests/test_oracle_golden.py: atol 5e-4) = 3.5e-3 mel units.
The bar here is that same 3.5e-3 mel units max-abs PLUS 1e-4 rel-L2 (measured: <= 2.2e-3 / <= 1.2e-5)."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden_drift as GD
from oracle import make_golden_full as GF
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig, TACOTRON_MEL_MAX, TACOTRON_MEL_MIN
from tortoise_tts_amd.schedule import Schedule
from tests.gpu_util import rel_err, max_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEL_ABS = 5e-4 * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) / 2  # 5e-4 in x units, in mel units


def gold(name):
    return np.load(os.path.join(GOLD, name))


def denorm(x):
    return (x + 1) / 2 * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN


@pytest.fixture(scope="module")
def sds():
    import bench
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    return bench.synthetic_weights()


def check(label, got, want, abs_tol=None, rel_tol=None):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    r, m = rel_err(got, want), max_err(got, want)
    print(f"[parity] FP32 MODE {label}: rel_l2={r:.3e} max_abs={m:.3e} (tol" + (f" abs {abs_tol:.0e}" if abs_tol else "") + (f" rel_l2 {rel_tol:.0e}" if rel_tol else "") + ")")
    assert torch.isfinite(got).all()
    if abs_tol is not None:
        assert m <= abs_tol, f"{label}: max_abs {m:.3e} > {abs_tol:.0e}"
    if rel_tol is not None:
        assert r <= rel_tol, f"{label}: rel_l2 {r:.3e} > {rel_tol:.0e}"


@torch.no_grad()
def test_fp32_ar_logits_and_latents(sds):
    """AR logits <= 1e-3 abs (prefill + KV-cached steps, teacher-forced) at the reference's default batch 16 and at the
    engine's 256; the latent re-pass of a winner."""
    g = gold("full_ar.npz")
    cfg = ARConfig()
    text, auto, _ = GF.prompt()
    toks = GF.ar_tokens()
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False  # suppressed (-1e9) in the benchmark weights
    st = stages.ArStage(sds["autoregressive"], cfg, dtype=E.TT_F32, max_batch=256, max_text=80, max_new_tokens=GF.LAT_N + 8, max_latent_candidates=1)
    for B in (GF.AR_B, 256):
        rep = B // GF.AR_B
        st.prefill(auto, text)
        check(f"AR prefill logits B={B}", st.logits(1)[0, keep], torch.from_numpy(g["logits"][0][0])[keep], abs_tol=1e-3)
        st.begin(B)
        for s, tk in enumerate(toks):
            st.decode_step(tk.repeat(rep))
            check(f"AR cached step {s + 1} logits B={B}", st.logits(B)[:, keep], torch.from_numpy(g["logits"][s + 1]).repeat(rep, 1)[:, keep], abs_tol=1e-3)
    check("AR latents", st.latents(auto, text, GF.latent_codes()), torch.from_numpy(g["latents"]), abs_tol=1e-3)
    assert st.guard() == 0
    st.close()


@torch.no_grad()
def test_fp32_clvp_scores(sds):
    cfg = CLVPConfig()
    text, _, _ = GF.prompt()
    codes = GF.clvp_codes()
    want = torch.from_numpy(gold("full_clvp.npz")["scores"])
    st = stages.ClvpStage(sds["clvp"], cfg, dtype=E.TT_F32, max_rows=8 * GF.CLVP_N)
    check("CLVP scores (768 / 12 / 20, 200 codes)", st.score(text, codes), want, abs_tol=1e-3 * float(want.abs().max()))
    st.close()


@torch.no_grad()
def test_fp32_diffusion_eps_and_schedules(sds):
    """eps <= 1e-4 rel-L2 per step; the final mel of the REAL schedules ('standard' 200 iterations and 'ultra_fast' 30
    iterations without guidance) <= 1e-3 abs against the reference's own fp32 p_sample_loop."""
    g = gold("full_diffusion.npz")
    cfg = DiffusionConfig()
    _, _, cond = GF.prompt()
    S, latents, x, step_noise = GF.diff_inputs(cfg)
    st = stages.DiffusionStage(sds["diffusion"], cfg, dtype=E.TT_F32, max_seq=S + 8, max_codes=GF.DIFF_M + 8, max_steps=200)
    st.condition(latents, cond, S)
    check("diffusion code_emb S=870", st.code_emb()[:, :, ::GF.CODE_EMB_STRIDE], torch.from_numpy(g["code_emb_strided"]), rel_tol=1e-4)
    out = st.forward(x, GF.DIFF_TS, cond_free=True)
    check("diffusion eps cond S=870", out[0], torch.from_numpy(g["eps_cond"])[0], rel_tol=1e-4)
    check("diffusion eps uncond S=870", out[1], torch.from_numpy(g["eps_uncond"])[0], rel_tol=1e-4)
    mel = st.sample(Schedule(GF.DIFF_LOOP_STEPS, 4000, True, 2.0), x, step_noise)
    check(f"diffusion {GF.DIFF_LOOP_STEPS}-step p_sample_loop mel", mel, denorm(torch.from_numpy(g["x0"])), abs_tol=MEL_ABS, rel_tol=1e-4)
    drift = np.load(os.path.join(GOLD, "full_drift.npz"))
    for case, N, cond_free, seed in GD.CASES:
        if case not in drift.files or N > 200:
            continue
        S2, lat2, x2, noise2 = GF.diff_inputs(cfg, M=GF.DIFF_M, seed=seed, steps=N)
        st.condition(lat2, cond, S2)
        mel = st.sample(Schedule(N, 4000, cond_free, 2.0), x2, noise2)
        check(f"diffusion {N}-iteration schedule (cond_free={cond_free}) final mel", mel, denorm(torch.from_numpy(drift[case])), abs_tol=MEL_ABS, rel_tol=1e-4)
    assert st.guard() == 0
    st.close()


@torch.no_grad()
def test_fp32_vocoder_waveform(sds):
    mel, z = GF.voc_inputs()
    st = stages.VocoderStage(sds["vocoder"], VocoderConfig(), dtype=E.TT_F32, max_frames=GF.VOC_S + 16)
    check("UnivNet waveform, 870 frames", st.inference(mel, z), torch.from_numpy(gold("full_vocoder.npz")["wav"]), abs_tol=1e-3)
    st.close()


@torch.no_grad()
def test_fp32_small_stages_vs_oracle():
    """Reduced-width stages end to end in the fp32 mode against the CPU oracle (same weights, no rounding anywhere): sampling loop codes
    bit-equal the oracle loop's on injected noise; conv taps / dilation / second activation source / split-K / statistics epilogues of
    the fp32 GEMM are all on this path."""
    from oracle import make_golden as G
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd import weights as W
    cfg = ARConfig(**G.AR_CFG)
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED)
    cond, text = G.ar_inputs(cfg)
    st = stages.ArStage(sd, cfg, dtype=E.TT_F32, max_batch=8, max_text=40, max_new_tokens=32, max_latent_candidates=2)
    st.prefill(cond, text)
    lg, kv = O.ar_prefill(sd, cfg, O.ar_prefix(sd, cfg, cond, text), G.AR_B)
    check("small AR prefill logits vs oracle", st.logits(1)[0], lg[0], abs_tol=1e-3)
    B, steps = 4, 16
    noise = torch.empty(steps, B, cfg.number_mel_codes).exponential_(1, generator=torch.Generator().manual_seed(3))
    want = O.ar_sample_loop(sd, cfg, cond, text, B, steps, noise)
    st.prefill(cond, text)
    got, n = st.generate(B, steps, exp_noise=noise)
    agree = float((got.cpu()[:, :want.shape[1]] == want[:, :got.shape[1]]).float().mean())
    print(f"[parity] FP32 MODE small AR sampling loop: {agree:.3f} of the codes equal the oracle loop's")
    assert agree >= 0.98
    st.close()
    dcfg = DiffusionConfig(**G.DIFF_CFG)
    dsd = W.synthetic_state_dict(W.diffusion_manifest(dcfg), seed=G.DIFF_SEED)
    S, latents, dcond, x, step_noise = G.diff_inputs(dcfg)
    df = stages.DiffusionStage(dsd, dcfg, dtype=E.TT_F32, max_seq=128, max_codes=64, max_steps=16)
    df.condition(latents, dcond, S)
    emb = O.diffusion_timestep_independent(dsd, dcfg, latents, dcond, S)
    check("small diffusion code_emb vs oracle", df.code_emb(), emb, rel_tol=1e-5)
    mel = df.sample(Schedule(G.DIFF_STEPS, 4000, True, 2.0), x, step_noise)
    want = O.denormalize_tacotron_mel(O.p_sample_loop(dsd, dcfg, O.Schedule(G.DIFF_STEPS, 4000, True, 2.0), emb, x.clone(), step_noise))
    check("small diffusion p_sample_loop mel vs oracle", mel, want, abs_tol=MEL_ABS, rel_tol=1e-4)
    df.close()
