"""TEST INFRASTRUCTURE ONLY — BASELINE-size golden vectors from the REFERENCE's own nn.Modules.

Run in the build container (needs /root/reference, ~2 minutes of CPU):   python -m oracle.make_golden_full
Writes tests/golden/full_*.npz: outputs of UnifiedVoice / GPT2InferenceModel, CLVP, DiffusionTts + SpacedDiffusion and
UnivNetGenerator built at the api.py:217-236 hyper-parameters with exactly the seeded synthetic weights and prompt
`bench.py` times (bench.synthetic_weights / synthetic_prompt), at the benchmarked shapes: AR batch 16 (the reference's
own default batch, api.py:156-157) with the 55-token prompt, 200 mel codes -> S = 870 denoiser positions, 768-wide
12-head 20-layer CLVP towers, 870 vocoder frames.  Only inputs that cannot be regenerated from a seed are stored;
weights are rebuilt bit-identically from (manifest, seed) on any machine with the same torch build.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from oracle import make_golden as G  # noqa: E402
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig  # noqa: E402

OUT = G.OUT

# shared case definitions (the GPU tests rebuild the inputs from these)
AR_B, AR_STEPS, AR_TOK_SEED = 16, 3, 21
LAT_N, LAT_SEED = 200, 22
CLVP_B, CLVP_N, CLVP_SEED = 4, 200, 23
DIFF_M, DIFF_SEED, DIFF_TS, DIFF_LOOP_STEPS = 200, 24, 2000, 8
CLVP64_B, CLVP64_SEED = 64, 27             # round 5: ranking of many candidates (Spearman / top-k agreement of the 16-bit engines)
VOC_S, VOC_SEED = 870, 25
DRIFT_STEPS, DRIFT_SEED = 200, 26       # 'standard' schedule length on the reduced-width denoiser (G.DIFF_CFG)
CODE_EMB_STRIDE = 8                     # code_emb is stored at every 8th position (3.5 MB otherwise)
# round 6: the decode step at the CONTEXTS the benchmark runs (1 .. 200 own keys; 500 = the maximum decode length): 8 distinct rows
# teacher-forced for 500 steps through the reference's GPT2InferenceModel; logits kept after feeding ARL_CHECK[i] tokens (= that many own
# keys in the cache: 63 / 64 / 65 and 127 / 128 straddle the 64-key blocks of the decode attention kernels), rows 0..3 only beyond 200
ARL_B, ARL_STEPS, ARL_TOK_SEED = 8, 500, 31
ARL_CHECK = (1, 63, 64, 65, 127, 128, 199, 200, 320, 499, 500)
ARL_ROWS_LONG = 4


def prompt():
    import bench
    text, (auto, diff) = bench.synthetic_prompt()
    return F.pad(text.int()[None], (0, 1)), auto, diff   # api.py:391 pad -> T = 55


def ar_tokens():
    g = torch.Generator().manual_seed(AR_TOK_SEED)
    return torch.randint(0, 8192, (AR_STEPS, AR_B), generator=g)


def arl_tokens():
    g = torch.Generator().manual_seed(ARL_TOK_SEED)
    return torch.randint(0, 8192, (ARL_STEPS, ARL_B), generator=g)


def latent_codes():
    g = torch.Generator().manual_seed(LAT_SEED)
    return torch.randint(0, 8192, (1, LAT_N), generator=g)


def clvp_codes():
    g = torch.Generator().manual_seed(CLVP_SEED)
    return torch.randint(0, 8192, (CLVP_B, CLVP_N), generator=g)


def clvp64_codes():
    g = torch.Generator().manual_seed(CLVP64_SEED)
    return torch.randint(0, 8192, (CLVP64_B, CLVP_N), generator=g)


def diff_inputs(cfg, M=DIFF_M, seed=DIFF_SEED, steps=DIFF_LOOP_STEPS):
    g = torch.Generator().manual_seed(seed)
    S = M * 4 * 24000 // 22050
    latents = torch.randn(1, M, cfg.in_latent_channels, generator=g)
    x = torch.randn(1, 100, S, generator=g)
    step_noise = torch.randn(steps, 1, 100, S, generator=g)
    return S, latents, x, step_noise


def voc_inputs():
    g = torch.Generator().manual_seed(VOC_SEED)
    mel = torch.randn(1, 100, VOC_S, generator=g) * 2 - 5
    z = torch.randn(1, 64, VOC_S + 10, generator=g)
    return mel, z


def build_ref_diffusion(ref, cfg, sd):
    m = ref.DiffusionTts(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
                         out_channels=cfg.out_channels, in_latent_channels=cfg.in_latent_channels, in_tokens=cfg.in_tokens,
                         dropout=0, use_fp16=False, num_heads=cfg.num_heads, layer_drop=0, unconditioned_percentage=0).eval()
    m.load_state_dict(sd, strict=True)
    return m


def ref_p_sample_loop(ref, m, N, S, x, code_emb, step_noise):
    """The reference's own SpacedDiffusion.p_sample_loop (utils/diffusion.py:533-621) with injected noise."""
    diffuser = ref.SpacedDiffusion(use_timesteps=ref.space_timesteps(4000, [N]), model_mean_type='epsilon',
                                   model_var_type='learned_range', loss_type='mse',
                                   betas=ref.get_named_beta_schedule('linear', 4000), conditioning_free=True,
                                   conditioning_free_k=2.0)
    import tortoise.utils.diffusion as rd
    order = list(reversed(range(N)))
    calls = {"n": 0}
    orig = rd.th.randn_like

    def fake_randn_like(t):
        i = order[calls["n"]]
        calls["n"] += 1
        return step_noise[i]
    rd.th.randn_like = fake_randn_like
    try:
        return diffuser.p_sample_loop(m, (1, 100, S), noise=x.clone(), model_kwargs={'precomputed_aligned_embeddings': code_emb},
                                      progress=False)
    finally:
        rd.th.randn_like = orig


@torch.no_grad()
def full_ar(ref, sds):
    cfg = ARConfig()
    m = G.build_ref_ar(ref, cfg, sds["autoregressive"])
    text, auto, _ = prompt()
    t = F.pad(text, (0, 1), value=m.stop_text_token)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([auto.unsqueeze(1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]
    ids = torch.full((AR_B, P + 1), 1, dtype=torch.long)
    ids[:, -1] = m.start_mel_token
    out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
    logits = [out.logits[:, -1]]
    past = out.past_key_values
    for tk in ar_tokens():
        ids = torch.cat([ids, tk[:, None]], dim=1)
        out = m.inference_model(input_ids=tk[:, None], past_key_values=past, attention_mask=torch.ones_like(ids),
                                use_cache=True, return_dict=True)
        past = out.past_key_values
        logits.append(out.logits[:, -1])
    codes = latent_codes()
    lat = m(auto, text, torch.tensor([text.shape[-1]]), codes.clone(), torch.tensor([LAT_N * m.mel_length_compression]),
            return_latent=True, clip_inputs=False)
    np.savez_compressed(os.path.join(OUT, "full_ar.npz"), logits=torch.stack(logits).numpy(), latents=lat.numpy())


@torch.no_grad()
def full_ar_long(ref, sds):
    """GPT2InferenceModel.forward (autoregressive.py:108-186) KV-cached for 500 teacher-forced steps: what the decode step computes at the
    contexts `bench.py` runs it at (own keys 1 .. 200) and at the maximum decode length (500, api.py:338 max_mel_tokens)."""
    cfg = ARConfig()
    m = G.build_ref_ar(ref, cfg, sds["autoregressive"])
    text, auto, _ = prompt()
    t = F.pad(text, (0, 1), value=m.stop_text_token)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([auto.unsqueeze(1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]
    ids = torch.full((ARL_B, P + 1), 1, dtype=torch.long)
    ids[:, -1] = m.start_mel_token
    out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
    past = out.past_key_values
    keep = {}
    for s, tk in enumerate(arl_tokens()):
        ids = torch.cat([ids, tk[:, None]], dim=1)
        out = m.inference_model(input_ids=tk[:, None], past_key_values=past, attention_mask=torch.ones_like(ids),
                                use_cache=True, return_dict=True)
        past = out.past_key_values
        if s + 1 in ARL_CHECK:
            rows = ARL_B if s + 1 <= 200 else ARL_ROWS_LONG
            keep["logits_%d" % (s + 1)] = out.logits[:rows, -1].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "full_ar_long.npz"), **keep)


@torch.no_grad()
def full_clvp(ref, sds):
    cfg = CLVPConfig()
    m = ref.CLVP(dim_text=cfg.dim, dim_speech=cfg.dim, dim_latent=cfg.dim_latent, num_text_tokens=256,
                 text_enc_depth=cfg.depth, text_seq_len=350, text_heads=cfg.heads, num_speech_tokens=8192,
                 speech_enc_depth=cfg.depth, speech_heads=cfg.heads, speech_seq_len=430, use_xformers=True).eval()
    m.load_state_dict(sds["clvp"], strict=True)
    text, _, _ = prompt()
    codes = clvp_codes()
    scores = m(text.long().repeat(CLVP_B, 1), codes, return_loss=False)  # api.py:463
    np.savez_compressed(os.path.join(OUT, "full_clvp.npz"), scores=scores.numpy())


@torch.no_grad()
def full_clvp64(ref, sds):
    """64 candidates for ONE text (api.py:460-477 ranks num_autoregressive_samples of them): the reference module's scores, against which
    the 16-bit engines' RANKING is held (tests/test_gpu_r5.py)."""
    cfg = CLVPConfig()
    m = ref.CLVP(dim_text=cfg.dim, dim_speech=cfg.dim, dim_latent=cfg.dim_latent, num_text_tokens=256,
                 text_enc_depth=cfg.depth, text_seq_len=350, text_heads=cfg.heads, num_speech_tokens=8192,
                 speech_enc_depth=cfg.depth, speech_heads=cfg.heads, speech_seq_len=430, use_xformers=True).eval()
    m.load_state_dict(sds["clvp"], strict=True)
    text, _, _ = prompt()
    codes = clvp64_codes()
    scores = m(text.long().repeat(CLVP64_B, 1), codes, return_loss=False)
    np.savez_compressed(os.path.join(OUT, "full_clvp64.npz"), scores=scores.numpy())


@torch.no_grad()
def full_diffusion(ref, sds):
    cfg = DiffusionConfig()
    m = build_ref_diffusion(ref, cfg, sds["diffusion"])
    _, _, cond = prompt()
    S, latents, x, step_noise = diff_inputs(cfg)
    code_emb = m.timestep_independent(latents, cond, S, False)
    ts = torch.tensor([DIFF_TS])
    eps_c = m(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=False)
    eps_u = m(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=True)
    x0 = ref_p_sample_loop(ref, m, DIFF_LOOP_STEPS, S, x, code_emb, step_noise)
    # bf16 / fp16 drift over the REAL 'standard' schedule length (200 iterations, cond_free) on the reduced-width denoiser
    from tortoise_tts_amd import weights as W
    dcfg = DiffusionConfig(**G.DIFF_CFG)
    dsd = W.synthetic_state_dict(W.diffusion_manifest(dcfg), seed=G.DIFF_SEED)
    dm = build_ref_diffusion(ref, dcfg, dsd)
    dS, dlat, dx, dnoise = diff_inputs(dcfg, M=G.DIFF_M, seed=DRIFT_SEED, steps=DRIFT_STEPS)
    g = torch.Generator().manual_seed(DRIFT_SEED + 1)
    dcond = torch.randn(1, 2 * dcfg.model_channels, generator=g)
    demb = dm.timestep_independent(dlat, dcond, dS, False)
    dx0 = ref_p_sample_loop(ref, dm, DRIFT_STEPS, dS, dx, demb, dnoise)
    np.savez_compressed(os.path.join(OUT, "full_diffusion.npz"), code_emb_strided=code_emb[:, :, ::CODE_EMB_STRIDE].numpy(),
                        eps_cond=eps_c.numpy(), eps_uncond=eps_u.numpy(), x0=x0.numpy(), drift_x0=dx0.numpy(), drift_cond=dcond.numpy())


@torch.no_grad()
def full_vocoder(ref, sds, seed=1234):
    from tortoise_tts_amd import weights as W
    raw = W.synthetic_state_dict(W.vocoder_manifest(VocoderConfig()), seed + 3)  # bench.synthetic_weights() before folding
    m = ref.UnivNetGenerator()
    m.load_state_dict(raw, strict=True)
    m.eval(inference=True)  # bakes weight norm in (vocoder.py:284-298)
    for k, v in m.state_dict().items():
        assert torch.allclose(sds["vocoder"][k], v, atol=1e-6), k  # == the folded weights the engine gets
    mel, z = voc_inputs()
    wav = m.inference(mel, z)
    np.savez_compressed(os.path.join(OUT, "full_vocoder.npz"), wav=wav.numpy())


def main():
    import bench
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shims.import_reference()
    sds = bench.synthetic_weights()
    if "--ar-long" in sys.argv:  # (round 6 addition: generate this file alone, ~2 minutes)
        full_ar_long(ref, sds)
        print("full_ar_long.npz", os.path.getsize(os.path.join(OUT, "full_ar_long.npz")))
        return
    if "--clvp64" in sys.argv:  # (round 5 addition: generate this file alone)
        full_clvp64(ref, sds)
        print("full_clvp64.npz", os.path.getsize(os.path.join(OUT, "full_clvp64.npz")))
        return
    full_ar(ref, sds)
    full_ar_long(ref, sds)
    full_clvp64(ref, sds)
    full_clvp(ref, sds)
    full_diffusion(ref, sds)
    full_vocoder(ref, sds)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("full_"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
