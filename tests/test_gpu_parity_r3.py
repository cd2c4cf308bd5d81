"""-m gpu: the parity cases the round-2 review asked for.

  (1) the EXACT computation `bench.py` times — full-width denoiser (1024 channels, 10 layers, S = 870, bench's synthetic weights and
      prompt) over the real 'standard' (200 iterations) and 'high_quality' (400 iterations, fp16) schedules with conditioning-free
      guidance, and the 'ultra_fast' schedule (30 iterations, cond_free=False) — against the final x0 of the reference's own
      SpacedDiffusion.p_sample_loop (tests/golden/full_drift.npz, oracle/make_golden_drift.py); rel-L2 AND max-abs asserted at
      twice the values measured on the MI355X (profiles/r03_parity_gpu.txt);
  (2) tt_diff_sample with cond_free=False against the oracle's loop (reduced width, every step);
  (3) the maximum decode length, M = 500 codes -> S = 2176 positions: flash attention at n = 2176, one denoiser call at 4352
      rows (a different tile plan from S = 870) and UnivNet at 2186 frames, against the oracle;
  (4) tts(noise_override=...) end to end on reduced-width engines against the composed oracle pipeline: same sampled codes,
      same CLVP winner, same calm-token trim, waveform within tolerance.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import make_golden as G
from oracle import make_golden_full as GF
from oracle import make_golden_drift as GD
from oracle import tortoise_oracle as O
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import (ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig, TACOTRON_MEL_MAX, TACOTRON_MEL_MIN)
from tortoise_tts_amd.schedule import Schedule
from tests.gpu_util import DTYPES, quantize_sd, report, rel_err, max_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def denorm(x):
    return (x + 1) / 2 * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN


@pytest.fixture(scope="module")
def sds():
    import bench
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    return bench.synthetic_weights()


# (case of oracle/make_golden_drift.py, operand type) -> (rel-L2 bound, max-abs bound in mel units) = 2 x measured on MI355X
DRIFT_BOUNDS = {  # measured (profiles/r03_parity_gpu.txt): std200 bf16 8.6e-3 / 1.13, f16 1.1e-3 / 0.115; hq400 bf16 8.6e-3 / 0.90, f16 1.1e-3 / 0.10;
    # uf30 bf16 4.8e-3 / 0.66, f16 5.7e-4 / 0.079
    ("std200", "bf16"): (1.8e-2, 2.3), ("std200", "f16"): (2.3e-3, 0.23),
    ("hq400", "f16"): (2.3e-3, 0.21), ("hq400", "bf16"): (1.8e-2, 1.9),
    ("uf30", "bf16"): (1.0e-2, 1.4), ("uf30", "f16"): (1.2e-3, 0.16),
    # round 6: BASELINE config #2's own schedule ('fast': 80 iterations, api.py:326); bounds = the 200-iteration ones until measured
    ("fast80", "bf16"): (1.8e-2, 2.3), ("fast80", "f16"): (2.3e-3, 0.23),
}


@pytest.mark.parametrize("case", [c[0] for c in GD.CASES])
@torch.no_grad()
def test_full_width_schedules_vs_reference_loop(sds, case):
    """Final mel of the benchmarked denoiser over the real schedules vs the reference's fp32 p_sample_loop."""
    path = os.path.join(GOLD, "full_drift.npz")
    g = np.load(path)
    if case not in g.files:
        pytest.skip(f"{case} not in full_drift.npz")
    _, N, cond_free, seed = [c for c in GD.CASES if c[0] == case][0]
    cfg = DiffusionConfig()
    _, _, cond = GF.prompt()
    S, latents, x, step_noise = GF.diff_inputs(cfg, M=GF.DIFF_M, seed=seed, steps=N)
    want = denorm(torch.from_numpy(g[case]))
    sched = Schedule(N, 4000, cond_free, 2.0)
    for name, dt, tdt, tol in DTYPES:
        st = stages.DiffusionStage(sds["diffusion"], cfg, dtype=dt, max_seq=S + 8, max_codes=GF.DIFF_M + 8, max_steps=N)
        st.condition(latents, cond, S)
        mel = st.sample(sched, x, step_noise).cpu()
        st.close()
        r, m = rel_err(mel, want), max_err(mel, want)
        rb, mb = DRIFT_BOUNDS[(case, name)]
        print(f"[parity] FULL-WIDTH {N}-iteration schedule (cond_free={cond_free}) S=870 {name} vs reference p_sample_loop: mel rel_l2={r:.3e} "
              f"max_abs={m:.3e} on [{float(want.min()):.2f}, {float(want.max()):.2f}] (bounds {rb:.1e} / {mb:.2f})")
        assert torch.isfinite(mel).all()
        assert r < rb and m < mb, f"{case} {name}: rel_l2 {r:.3e} (bound {rb:.1e}) max_abs {m:.3e} (bound {mb})"


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@torch.no_grad()
def test_diff_sample_without_conditioning_free_guidance(name, dt, tdt, tol):
    """'ultra_fast' (api.py:325) samples with cond_free=False: p_mean_variance takes the plain branch (diffusion.py:341-384), the
    denoiser is evaluated on ONE row per step and the sampler epilogue has no guidance blend."""
    cfg = DiffusionConfig(**G.DIFF_CFG)
    sd = quantize_sd(W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=G.DIFF_SEED), tdt)
    S, latents, cond, x, _ = G.diff_inputs(cfg)
    N = 12
    gen = torch.Generator().manual_seed(77)
    step_noise = torch.randn(N, 1, 100, S, generator=gen)
    st = stages.DiffusionStage(sd, cfg, dtype=dt, max_seq=128, max_codes=64, max_steps=16)
    st.condition(latents, cond, S)
    emb = O.diffusion_timestep_independent(sd, cfg, latents, cond, S)
    mel = st.sample(Schedule(N, 4000, False, 2.0), x, step_noise)
    want = O.denormalize_tacotron_mel(O.p_sample_loop(sd, cfg, O.Schedule(N, 4000, False, 2.0), emb, x.clone(), step_noise))
    report(f"diffusion p_sample_loop cond_free=False ({N} steps) {name} vs oracle", mel, want, tol * 2)
    # and it is NOT the guided result: the two branches must differ on the same inputs
    guided = st.sample(Schedule(N, 4000, True, 2.0), x, step_noise)
    assert rel_err(guided, mel) > 1e-3
    st.close()


def _attn_ref(q, k, v, relpos):
    w = q @ k.transpose(-1, -2)
    n = q.shape[-2]
    pos = torch.arange(n, device=q.device)
    d = (pos[None, :] - pos[:, None]).clamp(-64, 64) + 64
    w = w + relpos[:, d][None]
    return torch.softmax(w, dim=-1) @ v


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_flash_attention_at_the_maximum_length(name, dt, tdt, tol):
    """n = 2176 (500 mel codes): 34 key tiles per query block, relative-position bias saturated beyond +-64."""
    lib = E.init()
    n, B, H = 2176, 2, 4
    g = torch.Generator().manual_seed(n)
    q = (torch.randn(B, H, n, 64, generator=g) * 0.125 * 2).to(tdt).cuda()
    k = (torch.randn(B, H, n, 64, generator=g) * 2).to(tdt).cuda()
    v = torch.randn(B, H, n, 64, generator=g).to(tdt).cuda()
    vt = v.transpose(-1, -2).contiguous()
    relpos = torch.randn(H, 129, generator=g).cuda()
    out = torch.zeros(B, n, H * 64, device="cuda", dtype=tdt)
    E.check(lib.tt_op_flash_attention(dt, E.ptr(q), E.ptr(k), E.ptr(vt), E.ptr(out), B, H, n, n, 0, E.ptr(relpos), None))
    torch.cuda.synchronize()
    ref = _attn_ref(q.float(), k.float(), v.float(), relpos).permute(0, 2, 1, 3).reshape(B, n, H * 64)
    report(f"flash {name} relpos n={n} B={B}", out.float(), ref, tol)


@torch.no_grad()
def test_denoiser_and_vocoder_at_500_codes(sds):
    """M = 500 -> S = 2176: one conditioned + conditioning-free denoiser call at 4352 rows (pick_tile: 128x128 / 128x64 instead of
    the S = 870 plan) and UnivNet over 2186 frames, vs the oracle on operand-rounded weights (bf16; fp16 shares every kernel)."""
    name, dt, tdt, tol = DTYPES[0]
    cfg = DiffusionConfig()
    _, _, cond = GF.prompt()
    S, latents, x, _ = GF.diff_inputs(cfg, M=500, seed=51, steps=1)
    assert S == 2176
    st = stages.DiffusionStage(sds["diffusion"], cfg, dtype=dt, max_seq=S + 8, max_codes=508, max_steps=4)
    st.condition(latents, cond, S)
    out = st.forward(x, 1500, cond_free=True).cpu()
    st.close()
    sdq = quantize_sd(sds["diffusion"], tdt)
    oemb = O.diffusion_timestep_independent(sdq, cfg, latents, cond, S)
    ts = torch.tensor([1500])
    report(f"FULL diffusion eps cond S=2176 {name} vs oracle", out[0], O.diffusion_forward(sdq, cfg, x, ts, oemb, False)[0], tol)
    report(f"FULL diffusion eps uncond S=2176 {name} vs oracle", out[1], O.diffusion_forward(sdq, cfg, x, ts, oemb, True)[0], tol)
    g = torch.Generator().manual_seed(52)
    mel = torch.randn(1, 100, S, generator=g) * 2 - 5
    z = torch.randn(1, 64, S + 10, generator=g)
    vs = stages.VocoderStage(sds["vocoder"], VocoderConfig(), dtype=dt, max_frames=S + 16)
    wav = vs.inference(mel, z).cpu()
    vs.close()
    want = O.univnet_inference(quantize_sd(sds["vocoder"], tdt), VocoderConfig(), mel, z)
    assert wav.shape == want.shape == (1, 1, S * 256)
    report(f"FULL UnivNet 2186 frames {name} vs oracle", wav, want, tol * 2)


@torch.no_grad()
def test_tts_end_to_end_vs_composed_oracle_pipeline():
    """tts(noise_override=...) on reduced-width engines against the oracle stages composed exactly as api.py:407-559 composes the
    reference's: sampled codes (bit-exact), fix_autoregressive_output, CLVP winner, latent re-pass, calm-token trim, S, diffusion
    loop, UnivNet.  fp16 operands (the sampled codes must agree token for token for the rest to be comparable)."""
    from tortoise_tts_amd.api import TextToSpeech, fix_autoregressive_output, calm_trim_length
    tdt = torch.float16
    a_cfg, c_cfg, d_cfg, v_cfg = ARConfig(**G.AR_CFG), CLVPConfig(**G.CLVP_CFG), DiffusionConfig(**G.DIFF_CFG), VocoderConfig()
    sds = {"autoregressive": quantize_sd(G.sampling_state_dict(a_cfg, 4.0), tdt),   # stop token likely: every candidate stops early (ragged), the loop exits early
           "clvp": quantize_sd(W.synthetic_state_dict(W.clvp_manifest(c_cfg), seed=G.CLVP_SEED), tdt),
           "diffusion": quantize_sd(W.synthetic_state_dict(W.diffusion_manifest(d_cfg), seed=G.DIFF_SEED), tdt),
           "vocoder": quantize_sd(W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(v_cfg), seed=G.VOC_SEED)), tdt)}
    tts = TextToSpeech(half=True, kv_cache=True, state_dicts=sds, configs={"ar": a_cfg, "clvp": c_cfg, "diffusion": d_cfg, "vocoder": v_cfg},
                       max_candidates=8, max_mel_tokens=40, max_text_tokens=40)
    cond, text = G.ar_inputs(a_cfg)          # auto latent [1, D], text ids [1, T] (before the api's pad)
    gen = torch.Generator().manual_seed(97)  # (oracle: winner 3 of 6 by a 0.34 margin, calm-token trim 28 of 36, loop exit after 21 steps)
    dcond = torch.randn(1, 2 * d_cfg.model_channels, generator=gen) * 0.5
    N, M, steps = 6, 36, 7
    S_max = M * 4 * 24000 // 22050
    noise = {"exp_noise": torch.empty(M, N, a_cfg.number_mel_codes).exponential_(1, generator=gen)}
    text_ids = text[0].tolist()
    # ---- the oracle pipeline (api.py:391, 407-427, 447-477, 516-524, 547-559)
    t_pad = F.pad(text.int(), (0, 1))
    codes = O.ar_sample_loop(sds["autoregressive"], a_cfg, cond, t_pad, N, M, noise["exp_noise"], 2.0, 0.8, 50, 0.8, kv_cache=True)
    padded = F.pad(codes, (0, M - codes.shape[1]), value=a_cfg.stop_mel_token)
    fixed = torch.from_numpy(np.stack([O.fix_autoregressive_output(r.numpy().copy(), a_cfg.stop_mel_token) for r in padded]))
    scores = O.clvp_score(sds["clvp"], c_cfg, t_pad.long().repeat(N, 1), fixed)
    best = int(torch.sort(-scores.double(), stable=True).indices[0])
    lat = O.ar_latents(sds["autoregressive"], a_cfg, cond, t_pad, fixed[best:best + 1])
    cut = calm_trim_length(fixed[best])
    lat = lat[:, :cut]
    S = lat.shape[1] * 4 * 24000 // 22050
    noise["x_T"] = torch.randn(1, 100, S, generator=gen)
    noise["step_noise"] = torch.randn(steps, 1, 100, S, generator=gen)
    noise["z"] = torch.randn(1, v_cfg.noise_dim, S + 10, generator=gen)
    emb = O.diffusion_timestep_independent(sds["diffusion"], d_cfg, lat, dcond, S)
    mel = O.denormalize_tacotron_mel(O.p_sample_loop(sds["diffusion"], d_cfg, O.Schedule(steps, 4000, True, 2.0), emb, noise["x_T"].clone(),
                                                       noise["step_noise"]))
    want = O.univnet_inference(sds["vocoder"], v_cfg, mel, noise["z"])
    # ---- the engine through the drop-in surface
    wav = tts.tts(text_ids, conditioning_latents=(cond, dcond), k=1, num_autoregressive_samples=N, max_mel_tokens=M, diffusion_iterations=steps,
                  cond_free=True, cond_free_k=2.0, temperature=0.8, top_p=0.8, repetition_penalty=2.0, use_deterministic_seed=5,
                  noise_override=noise)
    got_best = tts.last_best_codes[0].cpu()
    print(f"[parity] END-TO-END tts(noise_override): oracle winner {best} of {N} (scores {[round(float(s), 4) for s in scores]}), "
          f"calm-trim {cut}/{M} latents, S={S} (max {S_max})")
    assert cut < M and codes.shape[1] < M, "the case lost its point: no calm-token trim / no early exit"
    assert torch.equal(got_best, fixed[best]), "the engine ranked a different candidate first (or sampled different codes)"
    assert wav.shape == want.shape == (1, 1, S * 256)
    report("END-TO-END tts(noise_override) waveform f16 vs composed oracle pipeline", wav, want, 4e-2)
    for s_ in (tts.ar, tts.clvp, tts.diffusion, tts.vocoder):
        s_.close()


def _utterances(cfg, Ms, steps, seed):
    g = torch.Generator().manual_seed(seed)
    items = []
    for M in Ms:
        S = M * 4 * 24000 // 22050
        items.append((torch.randn(1, M, cfg.in_latent_channels, generator=g), torch.randn(1, 2 * cfg.model_channels, generator=g) * 0.5, S,
                      torch.randn(1, 100, S, generator=g), torch.randn(steps, 1, 100, S, generator=g)))
    return items


@pytest.mark.parametrize("cond_free", [True, False])
@torch.no_grad()
def test_sample_many_treats_every_utterance_as_if_it_ran_alone(cond_free):
    """tt_diff_sample_batch (BASELINE config #4: long-form chunks "batched through DiffusionTts"; the reference renders them one
    after the other, read.py:66-71): three utterances of different lengths share every denoiser pass, padded to the longest.  Each
    must come out as from sample() alone - its own GroupNorm statistics, its own attention span, zero padding past its own end.
    Reduced width (statistics by the separate pass, whose row chunks are anchored at the sample start): bit-identical, as long as
    the batch does not change WHICH attention kernel an utterance gets (sequences of <= 128 positions run the register-prefetch
    flash kernel when alone and the LDS-staged one inside a longer batch: different key tiling, i.e. different bf16 rounding points
    of the online softmax - that case is held to the operand tolerance instead)."""
    cfg = DiffusionConfig(**G.DIFF_CFG)
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=G.DIFF_SEED)
    N = 6
    items = _utterances(cfg, (46, 31, 38), N, 5)
    sched = Schedule(N, 4000, cond_free, 2.0)
    st = stages.DiffusionStage(sd, cfg, max_seq=256, max_codes=64, max_steps=16, max_batch=4)
    alone = []
    for lat, cond, S, x, nz in items:
        st.condition(lat, cond, S)
        alone.append(st.sample(sched, x, nz).clone())
    many = st.sample_many(sched, items)
    for u, (a, b) in enumerate(zip(many, alone)):
        assert a.shape == b.shape and torch.isfinite(a).all()
        print(f"[parity] sample_many (cond_free={cond_free}) utterance {u} S={a.shape[-1]} vs sample() alone: rel_l2={rel_err(a, b):.3e} max_abs={max_err(a, b):.3e}")
        assert torch.equal(a, b), f"utterance {u} of the padded batch differs from the same utterance alone"
    # any order, any subset: an utterance does not depend on its neighbours
    again = st.sample_many(sched, [items[1], items[0]])
    assert torch.equal(again[0], alone[1]) and torch.equal(again[1], alone[0])
    # and the handle goes back to single-utterance work afterwards
    st.condition(*items[2][:3])
    assert torch.equal(st.sample(sched, items[2][3], items[2][4]), alone[2])
    # a short utterance (52 positions) next to a long one: other attention kernel than alone -> operand tolerance
    short = _utterances(cfg, (12,), N, 6)[0]
    st.condition(*short[:3])
    want = st.sample(sched, short[3], short[4]).clone()
    got = st.sample_many(sched, [items[0], short])
    assert torch.equal(got[0], alone[0])
    report(f"sample_many (cond_free={cond_free}) 52-position utterance beside a 200-position one vs alone", got[1], want, 2.5e-2)
    st.close()


@torch.no_grad()
def test_sample_many_full_width(sds):
    """The same at the benchmarked width (1024 channels, 10 layers), where the GroupNorm statistics ride the producing GEMM's
    epilogue in row-tile groups anchored at the start of the BATCH: the grouping of the f32 partial sums differs from the
    single-utterance run (not the set of values summed).  A perturbation of one ulp of f32 flips a few bf16 roundings of the next
    operand, and after three or four layers the rounding decisions of the two runs are uncorrelated - two equally valid
    evaluations that differ by the operand noise floor.  So the bar is the operand tolerance of each type (fp16: 8x tighter,
    which is what tells a masking error from rounding noise)."""
    cfg = DiffusionConfig()
    N = 5
    items = _utterances(cfg, (60, 35), N, 9)
    sched = Schedule(N, 4000, True, 2.0)
    for name, dt, tdt, tol in DTYPES:
        st = stages.DiffusionStage(sds["diffusion"], cfg, dtype=dt, max_seq=272, max_codes=64, max_steps=8, max_batch=2)
        alone = []
        for lat, cond, S, x, nz in items:
            st.condition(lat, cond, S)
            alone.append(st.sample(sched, x, nz).clone())
        many = st.sample_many(sched, items)
        for u, (a, b) in enumerate(zip(many, alone)):
            r, m = rel_err(a, b), max_err(a, b)
            print(f"[parity] FULL-WIDTH sample_many {name} utterance {u} S={a.shape[-1]} vs sample() alone: rel_l2={r:.3e} max_abs={m:.3e} (tol rel_l2 {tol:.1e})")
            assert torch.isfinite(a).all() and r < tol
        st.close()
