"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz from the REFERENCE's own nn.Modules.

Run in the build container (needs /root/reference):   python -m oracle.make_golden
The reference holds no numeric golden vectors, so these fixtures are outputs of the
reference's classes (UnifiedVoice / GPT2InferenceModel, CLVP, DiffusionTts + SpacedDiffusion,
UnivNetGenerator, api.fix_autoregressive_output) executed here on CPU in fp32 with the seeded
synthetic weights of tortoise_tts_amd.weights (regenerated bit-identically from (manifest, seed) on
any machine with the same torch build, so only inputs and outputs are stored).  The GPU box has no
/root/reference; tests there compare the oracle and the HIP engine against these files.
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shims
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, VocoderConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# shared case definitions (tests import these so inputs are rebuilt identically)
AR_CFG = dict(layers=2, model_dim=128, heads=2)
AR_SEED, AR_B, AR_T = 11, 3, 9
AR_TOKENS = [[5, 77, 8000], [1, 2, 3], [4000, 4001, 9]]
LAT_SEED, LAT_K, LAT_N, LAT_T = 12, 2, 24, 7
CLVP_CFG = dict(dim=128, dim_latent=128, depth=2, heads=2)
CLVP_SEED, CLVP_B, CLVP_T, CLVP_N = 13, 3, 13, 40
# CVVP (tts(cvvp_amount > 0), api.py:450-472): a reduced instance and the instance api.py:254 builds (512 wide, 8 heads, depth 8)
CVVP_CFG = dict(model_dim=128, heads=2, depth=2)
CVVP_SEED, CVVP_CLIPS, CVVP_T, CVVP_B, CVVP_N = 31, 2, 61, 5, 37
CVVP_FULL_CLIPS, CVVP_FULL_T, CVVP_FULL_B, CVVP_FULL_N = 2, 130, 6, 100
DIFF_CFG = dict(model_channels=128, num_layers=2, in_latent_channels=128, num_heads=2)
DIFF_SEED, DIFF_M, DIFF_STEPS, DIFF_TS = 14, 12, 5, 2999
VOC_SEED, VOC_S = 15, 6
COND_SEED, COND_CLIPS, COND_T_AR, COND_T_DIFF = 16, 2, 37, 44
# HF generate() cases pinning the sampling loop: (kv_cache, logit boost of the stop token); B rows x up to N tokens
SAMPLE_SEED, SAMPLE_B, SAMPLE_N = 5, 6, 24
SAMPLE_CASES = [(True, None), (True, 3.0), (True, 5.0), (False, None), (False, 3.0), (False, 5.0)]
# typical sampling (tts(typical_sampling=True, typical_mass=...), api.py:361-364): (kv_cache, stop-token boost, typical_mass) for the
# whole generate() loop, and warper-level vectors at the model's vocabulary: (seed, logit scale, typical_mass)
TYPICAL_CASES = [(True, None, 0.9), (True, 3.0, 0.5), (False, None, 0.2)]
TYPICAL_VOCAB, TYPICAL_ROWS = 8194, 4
TYPICAL_WARP_CASES = [(0, 1.0, 0.9), (1, 3.0, 0.9), (2, 6.0, 0.5), (3, 2.0, 0.2), (4, 0.25, 0.99)]


def ar_inputs(cfg):
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(1, cfg.model_dim, generator=g)
    text = F.pad(torch.randint(1, 255, (1, AR_T), generator=g).int(), (0, 1))
    return cond, text


def latent_inputs(cfg):
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(1, cfg.model_dim, generator=g)
    text = F.pad(torch.randint(1, 255, (1, LAT_T), generator=g).int(), (0, 1))
    codes = torch.randint(0, 8192, (LAT_K, LAT_N), generator=g)
    return cond, text, codes


def clvp_inputs():
    g = torch.Generator().manual_seed(2)
    text = torch.randint(0, 256, (1, CLVP_T), generator=g)
    codes = torch.randint(0, 8192, (CLVP_B, CLVP_N), generator=g)
    return text, codes


def cvvp_inputs(full=False):
    """auto_conds f32 [1, n_clips, 80, T] (the voice clips' mel spectrograms, api.py:262-276) and candidate codes int64 [B, n]."""
    clips, T, B, n = (CVVP_FULL_CLIPS, CVVP_FULL_T, CVVP_FULL_B, CVVP_FULL_N) if full else (CVVP_CLIPS, CVVP_T, CVVP_B, CVVP_N)
    g = torch.Generator().manual_seed(17 if full else 7)
    mels = torch.randn(1, clips, 80, T, generator=g) * 2 - 5
    codes = torch.randint(0, 8192, (B, n), generator=g)
    return mels, codes


def diff_inputs(cfg):
    g = torch.Generator().manual_seed(3)
    S = DIFF_M * 4 * 24000 // 22050
    latents = torch.randn(1, DIFF_M, cfg.in_latent_channels, generator=g)
    cond = torch.randn(1, 2 * cfg.model_channels, generator=g)
    x = torch.randn(1, 100, S, generator=g)
    step_noise = torch.randn(DIFF_STEPS, 1, 100, S, generator=g)
    return S, latents, cond, x, step_noise


def cond_inputs():
    """Two voice clips as mel spectrograms for both conditioning encoders (api.py:258-299 builds these from audio)."""
    g = torch.Generator().manual_seed(COND_SEED)
    return (torch.randn(1, COND_CLIPS, 80, COND_T_AR, generator=g), torch.randn(1, COND_CLIPS, 100, COND_T_DIFF, generator=g))


def voc_inputs():
    g = torch.Generator().manual_seed(4)
    mel = torch.randn(1, 100, VOC_S, generator=g) * 2 - 5
    z = torch.randn(1, 64, VOC_S + 10, generator=g)
    return mel, z


def sampling_state_dict(cfg, eos_boost):
    """AR weights of the sampling cases: the small AR golden weights with the stop-token logit raised by `eos_boost`
    (None: untouched) so rows finish at different steps (ragged EOS) or the whole batch stops early."""
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=AR_SEED)
    if eos_boost is not None:
        b = sd["mel_head.bias"].clone()
        b[cfg.stop_mel_token] += eos_boost
        sd["mel_head.bias"] = b
    return sd


def sampling_noise(cfg):
    """Exp(1) draws [N, B, V] that torch.multinomial consumes under torch.manual_seed(SAMPLE_SEED): multinomial(p, 1) on
    CPU == argmax(p / q) with q = empty_like(p).exponential_() from the same generator state."""
    torch.manual_seed(SAMPLE_SEED)
    return torch.stack([torch.empty(SAMPLE_B, cfg.number_mel_codes).exponential_(1) for _ in range(SAMPLE_N)])


def build_ref_ar(ref, cfg, sd, kv_cache=True):
    m = ref.UnifiedVoice(max_mel_tokens=cfg.max_mel_tokens, max_text_tokens=cfg.max_text_tokens,
                         max_conditioning_inputs=cfg.max_conditioning_inputs, layers=cfg.layers, model_dim=cfg.model_dim,
                         heads=cfg.heads, number_text_tokens=cfg.number_text_tokens, start_text_token=cfg.start_text_token,
                         checkpointing=False, train_solo_embeddings=False).eval()
    m.load_state_dict(sd, strict=True)
    m.post_init_gpt2_config(kv_cache=kv_cache)
    return m


@torch.no_grad()
def hf_generate_codes(ref, cfg, sd, kv_cache, typical_mass=None):
    """Codes of a real HF `generate(do_sample=True, ...)` run of the reference model through its own
    UnifiedVoice.inference_speech (autoregressive.py:535-563) with the api.py:416-424 arguments, on CPU.
    typical_mass: additionally `typical_sampling=True, typical_mass=...` (tts()'s hf_generate_kwargs, api.py:361-364), which
    inference_speech turns into its logits_processor list [TypicalLogitsWarper] (autoregressive.py:558)."""
    m = ref_shims.enable_generate(build_ref_ar(ref, cfg, sd, kv_cache))
    cond, text = ar_inputs(cfg)
    torch.manual_seed(SAMPLE_SEED)
    extra = {} if typical_mass is None else {"typical_sampling": True, "typical_mass": typical_mass}
    return m.inference_speech(cond, text, do_sample=True, top_p=0.8, temperature=0.8, num_return_sequences=SAMPLE_B,
                              length_penalty=1, repetition_penalty=2.0, max_generate_length=SAMPLE_N, **extra)


@torch.no_grad()
def golden_sampling(ref):
    cfg = ARConfig(**AR_CFG)
    out = {}
    for kv, boost in SAMPLE_CASES:
        codes = hf_generate_codes(ref, cfg, sampling_state_dict(cfg, boost), kv)
        out[f"codes_kv{int(kv)}_eos{boost}"] = codes.numpy()
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **out)


def typical_warp_scores(seed, scale):
    """Scores [TYPICAL_ROWS, TYPICAL_VOCAB] as the typical warper sees them (after the repetition penalty): seeded normal logits,
    the stop token suppressed in every row (the benchmark's EOS suppression), a band of -inf in row 2, two exactly equal logits in row 3."""
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(TYPICAL_ROWS, TYPICAL_VOCAB, generator=g) * scale
    x[:, TYPICAL_VOCAB - 1] = -float("inf")
    x[2, 100:4000] = -float("inf")
    x[3, 17] = x[3, 4242]
    return x


@torch.no_grad()
def golden_typical(ref):
    """The reference's OWN TypicalLogitsWarper (tortoise/utils/typical_sampling.py) on fixed score rows, and real HF generate() runs of
    the reference model through inference_speech(typical_sampling=True, typical_mass=...) (autoregressive.py:535-563)."""
    from tortoise.utils.typical_sampling import TypicalLogitsWarper
    cfg = ARConfig(**AR_CFG)
    out = {}
    for seed, scale, mass in TYPICAL_WARP_CASES:
        kept = TypicalLogitsWarper(mass=mass)(None, typical_warp_scores(seed, scale)) > -float("inf")
        out[f"kept_s{seed}"] = np.packbits(kept.numpy(), axis=1)
    for kv, boost, mass in TYPICAL_CASES:
        codes = hf_generate_codes(ref, cfg, sampling_state_dict(cfg, boost), kv, typical_mass=mass)
        out[f"codes_kv{int(kv)}_eos{boost}_mass{mass}"] = codes.numpy()
    np.savez_compressed(os.path.join(OUT, "typical.npz"), **out)


@torch.no_grad()
def golden_ar(ref):
    cfg = ARConfig(**AR_CFG)
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=AR_SEED)
    m = build_ref_ar(ref, cfg, sd)
    cond, text = ar_inputs(cfg)
    t = F.pad(text, (0, 1), value=m.stop_text_token)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([cond.unsqueeze(1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]
    ids = torch.full((AR_B, P + 1), 1, dtype=torch.long)
    ids[:, -1] = m.start_mel_token
    out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
    logits = [out.logits[:, -1]]
    past = out.past_key_values
    for tk in AR_TOKENS:
        tk = torch.tensor(tk)
        ids = torch.cat([ids, tk[:, None]], dim=1)
        out = m.inference_model(input_ids=tk[:, None], past_key_values=past, attention_mask=torch.ones_like(ids),
                                use_cache=True, return_dict=True)
        past = out.past_key_values
        logits.append(out.logits[:, -1])
    # latent re-pass
    cfg2 = ARConfig(**AR_CFG)
    sd2 = W.synthetic_state_dict(W.ar_manifest(cfg2), seed=LAT_SEED)
    m2 = build_ref_ar(ref, cfg2, sd2)
    cond2, text2, codes2 = latent_inputs(cfg2)
    lat = m2(cond2.repeat(LAT_K, 1), text2.repeat(LAT_K, 1), torch.tensor([text2.shape[-1]]), codes2.clone(),
             torch.tensor([LAT_N * m2.mel_length_compression]), return_latent=True, clip_inputs=False)
    np.savez_compressed(os.path.join(OUT, "ar.npz"), prefix_emb=emb.numpy(), logits=torch.stack(logits).numpy(),
                        latents=lat.numpy())


@torch.no_grad()
def golden_clvp(ref):
    cfg = CLVPConfig(**CLVP_CFG)
    sd = W.synthetic_state_dict(W.clvp_manifest(cfg), seed=CLVP_SEED)
    m = ref.CLVP(dim_text=cfg.dim, dim_speech=cfg.dim, dim_latent=cfg.dim_latent, num_text_tokens=256,
                 text_enc_depth=cfg.depth, text_seq_len=350, text_heads=cfg.heads, num_speech_tokens=8192,
                 speech_enc_depth=cfg.depth, speech_heads=cfg.heads, speech_seq_len=430, use_xformers=True).eval()
    m.load_state_dict(sd, strict=True)
    text, codes = clvp_inputs()
    scores = m(text.repeat(CLVP_B, 1), codes, return_loss=False)
    np.savez_compressed(os.path.join(OUT, "clvp.npz"), scores=scores.numpy())


@torch.no_grad()
def golden_cvvp(ref):
    """The reference's CVVP class (tortoise/models/cvvp.py) as api.py:464-468 drives it: per conditioning clip, the clip repeated for
    every candidate; mean over the clips.  Reduced instance and the api.py:254 instance."""
    from tortoise.models.cvvp import CVVP
    out = {}
    for tag, cfg, full in (("small", CVVPConfig(**CVVP_CFG), False), ("full", CVVPConfig(), True)):
        sd = W.synthetic_state_dict(W.cvvp_manifest(cfg), seed=CVVP_SEED)
        m = CVVP(model_dim=cfg.model_dim, transformer_heads=cfg.heads, dropout=0, mel_codes=cfg.mel_codes, conditioning_enc_depth=cfg.depth,
                 cond_mask_percentage=0, speech_enc_depth=cfg.depth, speech_mask_percentage=0, latent_multiplier=cfg.latent_multiplier).eval()
        m.load_state_dict(sd, strict=True)
        mels, codes = cvvp_inputs(full)
        acc = 0
        for cl in range(mels.shape[1]):
            acc = acc + m(mels[:, cl].repeat(codes.shape[0], 1, 1), codes, return_loss=False)
        out[f"scores_{tag}"] = (acc / mels.shape[1]).numpy()
    np.savez_compressed(os.path.join(OUT, "cvvp.npz"), **out)


@torch.no_grad()
def golden_diffusion(ref):
    cfg = DiffusionConfig(**DIFF_CFG)
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=DIFF_SEED)
    m = ref.DiffusionTts(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
                         out_channels=cfg.out_channels, in_latent_channels=cfg.in_latent_channels, in_tokens=cfg.in_tokens,
                         dropout=0, use_fp16=False, num_heads=cfg.num_heads, layer_drop=0, unconditioned_percentage=0).eval()
    m.load_state_dict(sd, strict=True)
    S, latents, cond, x, step_noise = diff_inputs(cfg)
    code_emb = m.timestep_independent(latents, cond, S, False)
    ts = torch.tensor([DIFF_TS])
    eps_c = m(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=False)
    eps_u = m(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=True)
    N = DIFF_STEPS
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
        x0 = diffuser.p_sample_loop(m, (1, 100, S), noise=x.clone(), model_kwargs={'precomputed_aligned_embeddings': code_emb},
                                    progress=False)
    finally:
        rd.th.randn_like = orig
    np.savez_compressed(os.path.join(OUT, "diffusion.npz"), code_emb=code_emb.numpy(), eps_cond=eps_c.numpy(),
                        eps_uncond=eps_u.numpy(), x0=x0.numpy(), timestep_map=np.array(diffuser.timestep_map))


@torch.no_grad()
def golden_vocoder(ref):
    cfg = VocoderConfig()
    raw = W.synthetic_state_dict(W.vocoder_manifest(cfg), seed=VOC_SEED)
    m = ref.UnivNetGenerator()
    m.load_state_dict(raw, strict=True)
    m.eval(inference=True)
    mel, z = voc_inputs()
    wav = m.inference(mel, z)
    np.savez_compressed(os.path.join(OUT, "vocoder.npz"), wav=wav.numpy())


@torch.no_grad()
def golden_conditioning(ref):
    """get_conditioning of both models on the small AR / diffusion configurations."""
    a_cfg, d_cfg = ARConfig(**AR_CFG), DiffusionConfig(**DIFF_CFG)
    a_sd = W.synthetic_state_dict(W.ar_manifest(a_cfg), seed=COND_SEED)
    d_sd = W.synthetic_state_dict(W.diffusion_manifest(d_cfg), seed=COND_SEED + 1)
    m = build_ref_ar(ref, a_cfg, a_sd)
    d = ref.DiffusionTts(model_channels=d_cfg.model_channels, num_layers=d_cfg.num_layers, in_channels=d_cfg.in_channels,
                         out_channels=d_cfg.out_channels, in_latent_channels=d_cfg.in_latent_channels, in_tokens=d_cfg.in_tokens,
                         dropout=0, use_fp16=False, num_heads=d_cfg.num_heads, layer_drop=0, unconditioned_percentage=0).eval()
    d.load_state_dict(d_sd, strict=True)
    mel_ar, mel_diff = cond_inputs()
    np.savez_compressed(os.path.join(OUT, "conditioning.npz"), auto_latent=m.get_conditioning(mel_ar).numpy(),
                        diffusion_latent=d.get_conditioning(mel_diff).numpy())


RLG_SEED = 17


def rlg_inputs(channels):
    g = torch.Generator().manual_seed(RLG_SEED + channels)
    return torch.randn(1, channels, generator=g)


@torch.no_grad()
def golden_rlg():
    """get_random_conditioning_latents (api.py:301-309): the reference's RandomLatentConverter at both widths, with the
    Gaussian input its forward draws (random_latent_generator.py:50) pinned by seeding the global generator."""
    ref_shims.install()
    from tortoise.models.random_latent_generator import RandomLatentConverter
    out = {}
    for ch in (1024, 2048):
        sd = W.synthetic_state_dict(W.rlg_manifest(ch), seed=RLG_SEED, gain=3.0)
        m = RandomLatentConverter(ch).eval()
        m.load_state_dict(sd, strict=True)
        r = rlg_inputs(ch)
        orig = torch.randn
        torch.randn = lambda *a, **k: r.clone()
        try:
            out[f"latent_{ch}"] = m(torch.tensor([0.0])).numpy()
        finally:
            torch.randn = orig
    np.savez_compressed(os.path.join(OUT, "rlg.npz"), **out)


HIFI_CFG = dict(in_channels=128, cond_channels=128, upsample_initial_channel=256)
HIFI_SEED, HIFI_T = 21, 12


def hifi_inputs(cfg):
    g = torch.Generator().manual_seed(HIFI_SEED + 1)
    return torch.randn(1, HIFI_T, cfg.in_channels, generator=g), torch.randn(1, cfg.cond_channels, generator=g) * 0.5


@torch.no_grad()
def golden_hifigan():
    """HifiganGenerator.inference of the streaming path (hifigan_decoder.py:259-289, api_fast.py:222-225 structure) at a
    reduced width: channels 256 -> 128 -> 64 -> 32 -> 16, factors [8, 8, 2, 2], three ResBlock1 per stage."""
    ref_shims.install()
    from tortoise.models.hifigan_decoder import HifiganGenerator
    from tortoise_tts_amd.config import HifiganConfig
    cfg = HifiganConfig(**HIFI_CFG)
    sd = W.synthetic_state_dict(W.hifigan_manifest(cfg), seed=HIFI_SEED)
    m = HifiganGenerator(in_channels=cfg.in_channels, out_channels=1, resblock_type="1",
                         resblock_dilation_sizes=[list(cfg.resblock_dilation_sizes)] * len(cfg.resblock_kernel_sizes),
                         resblock_kernel_sizes=list(cfg.resblock_kernel_sizes), upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
                         upsample_initial_channel=cfg.upsample_initial_channel, upsample_factors=list(cfg.upsample_factors),
                         cond_channels=cfg.cond_channels).eval()
    m.load_state_dict(sd, strict=True)
    m.device = torch.device("cpu")
    lat, g = hifi_inputs(cfg)
    np.savez_compressed(os.path.join(OUT, "hifigan.npz"), wav=m.inference(lat, g).numpy())


def golden_text():
    """Long-form chunking (tortoise/utils/text.py:4-72).  The three cases are the reference's OWN expectations
    (text.py:82-130, which pass here: `python tortoise/utils/text.py`); inputs and outputs are stored so the GPU box,
    which has no reference tree, can check them too."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("_ref_text", os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "utils", "text.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "data", "riding_hood.txt")) as f:
        riding_hood = f.read()
    cases = [
        {"text": """
            This is a sample sentence.
            This is another sample sentence.
            This is a longer sample sentence that should force a split inthemiddlebutinotinthislongword.
            "Don't split my quote... please"
            """, "desired_length": 20, "max_length": 40},
        {"text": """
            When you are really angry sometimes you use consecutive exclamation marks!!!!!! Is this a good thing to do?!?!?!
            I don't know but we should handle this situation..........................
            """, "desired_length": 30, "max_length": 50},
        {"text": riding_hood, "desired_length": 200, "max_length": 300},
    ]
    for c in cases:
        c["chunks"] = mod.split_and_recombine_text(c["text"], c["desired_length"], c["max_length"])
    assert cases[0]["chunks"][2] == "This is a longer sample sentence that" and len(cases[2]["chunks"]) == 15
    with open(os.path.join(OUT, "text_split.json"), "w") as f:
        json.dump(cases, f, indent=1)


def golden_integer():
    src = open(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "api.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fix_autoregressive_output"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "api_excerpt", "exec"), ns)
    rng = np.random.default_rng(0)
    ins, outs = [], []
    for trial in range(40):
        n = 32
        codes = rng.integers(0, 8192, n)
        if trial % 3:
            codes[int(rng.integers(0, n)):] = 8193
        if trial % 7 == 0:
            codes[int(rng.integers(0, n))] = 8193
        ins.append(codes.copy())
        outs.append(ns["fix_autoregressive_output"](torch.tensor(codes).clone(), 8193, complain=False).numpy())
    np.savez_compressed(os.path.join(OUT, "integer.npz"), codes_in=np.stack(ins), codes_out=np.stack(outs))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shims.import_reference()
    golden_ar(ref)
    golden_sampling(ref)
    golden_typical(ref)
    golden_clvp(ref)
    golden_cvvp(ref)
    golden_diffusion(ref)
    golden_vocoder(ref)
    golden_conditioning(ref)
    golden_rlg()
    golden_hifigan()
    golden_text()
    golden_integer()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
