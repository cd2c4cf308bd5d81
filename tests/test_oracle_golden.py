"""Pins oracle/tortoise_oracle.py against the committed golden vectors (tests/golden/*.npz), which are
outputs of the reference's own nn.Modules (oracle/make_golden.py).  Runs anywhere (no reference tree,
no GPU needed)."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as G
from oracle import tortoise_oracle as O
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, VocoderConfig

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, atol, rtol=1e-4):
    a = torch.as_tensor(np.asarray(a))
    b = torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), float((a - b).abs().max())


@torch.no_grad()
def test_ar_logits_and_latents():
    g = gold("ar.npz")
    cfg = ARConfig(**G.AR_CFG)
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED)
    cond, text = G.ar_inputs(cfg)
    prefix = O.ar_prefix(sd, cfg, cond, text)
    close(prefix, g["prefix_emb"], 1e-6)
    lg, kv = O.ar_prefill(sd, cfg, prefix, G.AR_B)
    close(lg, g["logits"][0], 2e-4)
    for s, tk in enumerate(G.AR_TOKENS):
        lg, kv = O.ar_step(sd, cfg, torch.tensor(tk), s + 1, kv)
        close(lg, g["logits"][s + 1], 2e-4)
    sd2 = W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.LAT_SEED)
    cond2, text2, codes2 = G.latent_inputs(cfg)
    lat = O.ar_latents(sd2, cfg, cond2.repeat(G.LAT_K, 1), text2.repeat(G.LAT_K, 1), codes2)
    close(lat, g["latents"], 2e-4)


@pytest.mark.parametrize("seed,scale,mass", G.TYPICAL_WARP_CASES)
def test_typical_warper_equals_reference_golden(seed, scale, mass):
    """oracle.typical_ == the reference's own TypicalLogitsWarper (tortoise/utils/typical_sampling.py:11-33) on the committed rows
    (tests/golden/typical.npz, oracle/make_golden.py:golden_typical): the kept set, token for token - rows with the stop token
    suppressed, a band of -inf, an exact tie."""
    x = G.typical_warp_scores(seed, scale)
    want = np.unpackbits(gold("typical.npz")[f"kept_s{seed}"], axis=1)[:, :G.TYPICAL_VOCAB].astype(bool)
    got = (O.typical_(x, mass) > -float("inf")).numpy()
    assert np.array_equal(got, want)
    assert 1 <= got.sum(1).min() and got.sum(1).max() < G.TYPICAL_VOCAB - 1  # the warper really removes something, and never everything


@pytest.mark.parametrize("kv_cache,eos_boost,mass", G.TYPICAL_CASES)
@torch.no_grad()
def test_sampling_loop_with_typical_sampling_equals_hf_generate_golden(kv_cache, eos_boost, mass):
    """oracle.ar_sample_loop(typical_mass=...) == the committed codes of HF generate() run through the reference's
    inference_speech(typical_sampling=True, typical_mass=...) (autoregressive.py:558: the warper rides in generate()'s logits_processor
    list, i.e. after the repetition penalty and before temperature / top-k / top-p), bit for bit."""
    cfg = ARConfig(**G.AR_CFG)
    want = gold("typical.npz")[f"codes_kv{int(kv_cache)}_eos{eos_boost}_mass{mass}"]
    sd = G.sampling_state_dict(cfg, eos_boost)
    cond, text = G.ar_inputs(cfg)
    got = O.ar_sample_loop(sd, cfg, cond, text, G.SAMPLE_B, G.SAMPLE_N, G.sampling_noise(cfg), kv_cache=kv_cache, typical_mass=mass)
    assert got.shape == want.shape and np.array_equal(got.numpy(), want)
    plain = gold("sampling.npz")[f"codes_kv{int(kv_cache)}_eos{eos_boost}"]
    assert plain.shape != want.shape or not np.array_equal(plain, want)  # and the option is not a no-op on these cases


@pytest.mark.parametrize("kv_cache,eos_boost", G.SAMPLE_CASES)
@torch.no_grad()
def test_sampling_loop_equals_hf_generate_golden(kv_cache, eos_boost):
    """oracle.ar_sample_loop == the committed codes of HF generate() run on the reference model (tests/golden/sampling.npz,
    oracle/make_golden.py:golden_sampling), bit for bit; runs without the reference tree (GPU box included)."""
    cfg = ARConfig(**G.AR_CFG)
    want = gold("sampling.npz")[f"codes_kv{int(kv_cache)}_eos{eos_boost}"]
    sd = G.sampling_state_dict(cfg, eos_boost)
    cond, text = G.ar_inputs(cfg)
    got = O.ar_sample_loop(sd, cfg, cond, text, G.SAMPLE_B, G.SAMPLE_N, G.sampling_noise(cfg), kv_cache=kv_cache)
    assert got.shape == want.shape and np.array_equal(got.numpy(), want)


@torch.no_grad()
def test_clvp_scores():
    cfg = CLVPConfig(**G.CLVP_CFG)
    sd = W.synthetic_state_dict(W.clvp_manifest(cfg), seed=G.CLVP_SEED)
    text, codes = G.clvp_inputs()
    close(O.clvp_score(sd, cfg, text.repeat(G.CLVP_B, 1), codes), gold("clvp.npz")["scores"], 1e-4)


@pytest.mark.parametrize("tag", ["small", "full"])
@torch.no_grad()
def test_cvvp_scores(tag):
    """oracle.cvvp_score == the committed scores of the reference's CVVP class driven as api.py:464-468 drives it (tests/golden/cvvp.npz):
    a reduced instance and the 512-wide / 8-head / depth-8 instance api.py:254 builds."""
    cfg = CVVPConfig(**G.CVVP_CFG) if tag == "small" else CVVPConfig()
    sd = W.synthetic_state_dict(W.cvvp_manifest(cfg), seed=G.CVVP_SEED)
    mels, codes = G.cvvp_inputs(tag == "full")
    close(O.cvvp_score(sd, cfg, mels, codes), gold("cvvp.npz")[f"scores_{tag}"], 1e-5)


@torch.no_grad()
def test_diffusion_network_and_sampler():
    g = gold("diffusion.npz")
    cfg = DiffusionConfig(**G.DIFF_CFG)
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=G.DIFF_SEED)
    S, latents, cond, x, step_noise = G.diff_inputs(cfg)
    emb = O.diffusion_timestep_independent(sd, cfg, latents, cond, S)
    close(emb, g["code_emb"], 1e-4)
    ts = torch.tensor([G.DIFF_TS])
    close(O.diffusion_forward(sd, cfg, x, ts, emb, False), g["eps_cond"], 2e-4)
    close(O.diffusion_forward(sd, cfg, x, ts, emb, True), g["eps_uncond"], 2e-4)
    sched = O.Schedule(G.DIFF_STEPS, 4000, True, 2.0)
    assert list(sched.timestep_map) == list(g["timestep_map"])
    close(O.p_sample_loop(sd, cfg, sched, emb, x.clone(), step_noise), g["x0"], 5e-4, 1e-3)


@torch.no_grad()
def test_univnet_waveform():
    cfg = VocoderConfig()
    sd = W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(cfg), seed=G.VOC_SEED))
    mel, z = G.voc_inputs()
    wav = O.univnet_inference(sd, cfg, mel, z)
    close(wav, gold("vocoder.npz")["wav"], 1e-4)
    assert wav.shape == (1, 1, G.VOC_S * 256)  # the reference's own shape check (vocoder.py:318-324)


@torch.no_grad()
def test_conditioning_encoders():
    """get_conditioning of both models: the oracle side of the next row, pinned against the
    reference modules' outputs so the device path can be built against it."""
    a_cfg, d_cfg = ARConfig(**G.AR_CFG), DiffusionConfig(**G.DIFF_CFG)
    a_sd = W.synthetic_state_dict(W.ar_manifest(a_cfg), seed=G.COND_SEED)
    d_sd = W.synthetic_state_dict(W.diffusion_manifest(d_cfg), seed=G.COND_SEED + 1)
    mel_ar, mel_diff = G.cond_inputs()
    g = gold("conditioning.npz")
    close(O.ar_get_conditioning(a_sd, a_cfg, mel_ar), g["auto_latent"], 1e-4)
    close(O.diffusion_get_conditioning(d_sd, d_cfg, mel_diff), g["diffusion_latent"], 1e-4)
    # mean over clips: one clip repeated gives that clip's latent; clip order does not matter for the AR encoder
    one = O.ar_get_conditioning(a_sd, a_cfg, mel_ar[:, :1])
    assert torch.allclose(O.ar_get_conditioning(a_sd, a_cfg, mel_ar[:, :1].repeat(1, 3, 1, 1)), one, atol=1e-5)
    assert torch.allclose(O.ar_get_conditioning(a_sd, a_cfg, mel_ar.flip(1)), O.ar_get_conditioning(a_sd, a_cfg, mel_ar), atol=1e-5)


def test_fix_autoregressive_output_bit_exact():
    g = gold("integer.npz")
    for cin, cout in zip(g["codes_in"], g["codes_out"]):
        assert np.array_equal(O.fix_autoregressive_output(cin, 8193), cout)


@torch.no_grad()
def test_random_latent_converter():
    """oracle.random_latent_converter vs outputs of the reference's RandomLatentConverter (tests/golden/rlg.npz)."""
    g = gold("rlg.npz")
    for ch in (1024, 2048):
        sd = W.synthetic_state_dict(W.rlg_manifest(ch), seed=G.RLG_SEED, gain=3.0)
        got = O.random_latent_converter(sd, G.rlg_inputs(ch))
        assert got.shape == (1, ch) and float(got.abs().mean()) > 1e-3
        close(got, g[f"latent_{ch}"], 1e-5)


@torch.no_grad()
def test_hifigan_decoder():
    """oracle.hifigan_inference vs the reference HifiganGenerator.inference output (hifigan_decoder.py:259-289)."""
    from tortoise_tts_amd.config import HifiganConfig
    cfg = HifiganConfig(**G.HIFI_CFG)
    sd = W.fold_weight_norm(W.synthetic_state_dict(W.hifigan_manifest(cfg), seed=G.HIFI_SEED))
    lat, g = G.hifi_inputs(cfg)
    close(O.hifigan_inference(sd, cfg, lat, g), gold("hifigan.npz")["wav"], 1e-6)
