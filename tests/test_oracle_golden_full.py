"""Pins oracle/tortoise_oracle.py at the BENCHMARKED configuration (api.py:217-236 hyper-parameters, bench.py's synthetic
weights and prompt, AR batch 16, S = 870, 768/12/20 CLVP towers, 870 vocoder frames) against outputs of the reference's
own nn.Modules committed as tests/golden/full_*.npz (oracle/make_golden_full.py).  CPU only, ~1 minute."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as G
from oracle import make_golden_full as GF
from oracle import tortoise_oracle as O
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def sds():
    import bench
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    return bench.synthetic_weights()


@torch.no_grad()
def test_full_ar_logits_and_latents(sds):
    g = gold("full_ar.npz")
    cfg = ARConfig()
    sd = sds["autoregressive"]
    text, auto, _ = GF.prompt()
    prefix = O.ar_prefix(sd, cfg, auto, text)
    lg, kv = O.ar_prefill(sd, cfg, prefix, GF.AR_B)
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False  # the suppressed (-1e9) stop logit would dominate every norm
    assert rel(lg[:, keep], g["logits"][0][:, keep]) < 2e-5
    for s, tk in enumerate(GF.ar_tokens()):
        lg, kv = O.ar_step(sd, cfg, tk, s + 1, kv)
        assert rel(lg[:, keep], g["logits"][s + 1][:, keep]) < 2e-5
    lat = O.ar_latents(sd, cfg, auto, text, GF.latent_codes())
    assert rel(lat, g["latents"]) < 2e-5


@torch.no_grad()
def test_full_clvp_scores(sds):
    text, _, _ = GF.prompt()
    got = O.clvp_score(sds["clvp"], CLVPConfig(), text.long().repeat(GF.CLVP_B, 1), GF.clvp_codes())
    assert rel(got, gold("full_clvp.npz")["scores"]) < 1e-4


@torch.no_grad()
def test_full_diffusion_network_and_sampler(sds):
    g = gold("full_diffusion.npz")
    cfg = DiffusionConfig()
    sd = sds["diffusion"]
    _, _, cond = GF.prompt()
    S, latents, x, step_noise = GF.diff_inputs(cfg)
    emb = O.diffusion_timestep_independent(sd, cfg, latents, cond, S)
    assert rel(emb[:, :, ::GF.CODE_EMB_STRIDE], g["code_emb_strided"]) < 2e-5
    ts = torch.tensor([GF.DIFF_TS])
    assert rel(O.diffusion_forward(sd, cfg, x, ts, emb, False), g["eps_cond"]) < 5e-5
    assert rel(O.diffusion_forward(sd, cfg, x, ts, emb, True), g["eps_uncond"]) < 5e-5
    sched = O.Schedule(GF.DIFF_LOOP_STEPS, 4000, True, 2.0)
    x0 = O.p_sample_loop(sd, cfg, sched, emb, x.clone(), step_noise)
    assert rel(x0, g["x0"]) < 2e-4


@torch.no_grad()
def test_drift_case_200_steps_reduced_width():
    """The 200-iteration ('standard' length) sampler on the reduced-width denoiser: the fp32 oracle tracks the reference's
    own p_sample_loop; the GPU drift test measures bf16 / fp16 against this same trajectory."""
    g = gold("full_diffusion.npz")
    cfg = DiffusionConfig(**G.DIFF_CFG)
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=G.DIFF_SEED)
    S, lat, x, noise = GF.diff_inputs(cfg, M=G.DIFF_M, seed=GF.DRIFT_SEED, steps=GF.DRIFT_STEPS)
    emb = O.diffusion_timestep_independent(sd, cfg, lat, torch.as_tensor(g["drift_cond"]), S)
    sched = O.Schedule(GF.DRIFT_STEPS, 4000, True, 2.0)
    x0 = O.p_sample_loop(sd, cfg, sched, emb, x.clone(), noise)
    assert rel(x0, g["drift_x0"]) < 1e-3


@torch.no_grad()
def test_full_vocoder(sds):
    mel, z = GF.voc_inputs()
    got = O.univnet_inference(sds["vocoder"], VocoderConfig(), mel, z)
    want = torch.as_tensor(gold("full_vocoder.npz")["wav"])
    assert got.shape == want.shape == (1, 1, GF.VOC_S * 256)
    assert float((got - want).abs().max()) < 1e-4
