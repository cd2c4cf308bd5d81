"""Pins oracle/tortoise_oracle.py against the reference's OWN nn.Modules, run live on CPU.
Only runs where /root/reference exists (the build container); skipped on the GPU box."""
import numpy as np
import pytest
import torch

from oracle import ref_shims
from oracle import tortoise_oracle as O
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, DiffusionConfig, CLVPConfig, VocoderConfig

pytestmark = pytest.mark.skipif(not ref_shims.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_shims.import_reference()


def small_ar():
    return ARConfig(layers=2, model_dim=128, heads=2)


def build_ref_ar(ref, cfg, sd):
    m = ref.UnifiedVoice(max_mel_tokens=cfg.max_mel_tokens, max_text_tokens=cfg.max_text_tokens,
                         max_conditioning_inputs=cfg.max_conditioning_inputs, layers=cfg.layers,
                         model_dim=cfg.model_dim, heads=cfg.heads, number_text_tokens=cfg.number_text_tokens,
                         start_text_token=cfg.start_text_token, checkpointing=False,
                         train_solo_embeddings=False).eval()
    m.load_state_dict(sd, strict=True)
    m.post_init_gpt2_config(kv_cache=True)
    return m


@torch.no_grad()
def test_ar_prefill_and_cached_steps(ref):
    cfg = small_ar()
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=11)
    m = build_ref_ar(ref, cfg, sd)
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(1, cfg.model_dim, generator=g)
    text = torch.randint(1, 255, (1, 9), generator=g).int()
    text = torch.nn.functional.pad(text, (0, 1))
    B = 3
    # reference: same prefix code path as inference_speech (autoregressive.py:538-548)
    t = torch.nn.functional.pad(text, (0, 1), value=m.stop_text_token)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([cond.unsqueeze(1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]
    ids = torch.full((B, P + 1), 1, dtype=torch.long)
    ids[:, -1] = m.start_mel_token
    out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
    ref_logits = [out.logits[:, -1]]
    toks = [torch.tensor([5, 77, 8000]), torch.tensor([1, 2, 3]), torch.tensor([4000, 4001, 9])]
    past = out.past_key_values
    for s, tk in enumerate(toks):
        ids = torch.cat([ids, tk[:, None]], dim=1)
        out = m.inference_model(input_ids=tk[:, None], past_key_values=past,
                                attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
        past = out.past_key_values
        ref_logits.append(out.logits[:, -1])
    # oracle
    prefix = O.ar_prefix(sd, cfg, cond, text)
    assert torch.allclose(prefix, emb, atol=1e-6)
    lg, kv = O.ar_prefill(sd, cfg, prefix, B)
    mine = [lg]
    for s, tk in enumerate(toks):
        lg, kv = O.ar_step(sd, cfg, tk, s + 1, kv)
        mine.append(lg)
    for a, b in zip(ref_logits, mine):
        assert torch.allclose(a, b, atol=2e-4, rtol=1e-4), (a - b).abs().max()


@pytest.mark.parametrize("kv_cache,eos_boost", [(True, None), (True, 3.0), (True, 5.0), (False, None), (False, 3.0), (False, 5.0)])
@torch.no_grad()
def test_sampling_loop_equals_hf_generate(ref, kv_cache, eos_boost):
    """oracle.ar_sample_loop against a REAL `generate(do_sample=True, top_p, temperature,
    repetition_penalty, num_return_sequences)` run of the reference's GPT2InferenceModel, driven through the reference's
    own UnifiedVoice.inference_speech (autoregressive.py:535-563) with the installed transformers' GenerationMixin mixed
    back in (oracle/ref_shims.enable_generate).  Same generator state => identical codes, bit for bit, including rows
    that hit the stop token at different steps (pad rule), whole-batch early exit, and both position rules
    (kv_cache=True: mel positions 0,2,3,...; kv_cache=False: 0,1,2,..., autoregressive.py:134-149)."""
    from oracle import make_golden as G
    cfg = small_ar()
    sd = G.sampling_state_dict(cfg, eos_boost)
    want = G.hf_generate_codes(ref, cfg, sd, kv_cache)
    cond, text = G.ar_inputs(cfg)
    got = O.ar_sample_loop(sd, cfg, cond, text, G.SAMPLE_B, G.SAMPLE_N, G.sampling_noise(cfg), kv_cache=kv_cache)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.equal(got, want)
    if eos_boost == 3.0:
        stops = (want == cfg.stop_mel_token).sum(1)
        assert stops.max() == G.SAMPLE_N and stops.min() == 0  # ragged: one row finished at step 0, others never
    if eos_boost == 5.0:
        assert want.shape[1] < G.SAMPLE_N  # every row finished: generate() returned early


@torch.no_grad()
def test_typical_sampling_equals_reference_warper_and_hf_generate(ref):
    """tts(typical_sampling=True, typical_mass=...) (api.py:361-364): (1) oracle.typical_ against the reference's OWN
    TypicalLogitsWarper (tortoise/utils/typical_sampling.py) live, on fresh rows and masses; (2) oracle.ar_sample_loop(typical_mass)
    against a real generate() run through the reference's inference_speech(typical_sampling=True) - same generator state =>
    identical codes."""
    from tortoise.utils.typical_sampling import TypicalLogitsWarper
    from oracle import make_golden as G
    for seed in range(3):
        g = torch.Generator().manual_seed(900 + seed)
        x = torch.randn(5, 8194, generator=g) * (0.5 + 2 * seed)
        x[:, 8193] = -float("inf")
        x[1, 50:5000] = -float("inf")
        for mass in (0.9, 0.6, 0.3):
            assert torch.equal(O.typical_(x, mass), TypicalLogitsWarper(mass=mass)(None, x))
    cfg = small_ar()
    kv_cache, eos_boost, mass = G.TYPICAL_CASES[1]
    sd = G.sampling_state_dict(cfg, eos_boost)
    want = G.hf_generate_codes(ref, cfg, sd, kv_cache, typical_mass=mass)
    cond, text = G.ar_inputs(cfg)
    got = O.ar_sample_loop(sd, cfg, cond, text, G.SAMPLE_B, G.SAMPLE_N, G.sampling_noise(cfg), kv_cache=kv_cache, typical_mass=mass)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("kv_cache,eos_boost", [(True, None), (True, 3.0), (True, 5.0), (False, None), (False, 3.0), (False, 5.0)])
@torch.no_grad()
def test_sampling_loop_and_streamed_latents_equal_reference_sample_stream(ref, kv_cache, eos_boost):
    """the reference's OWN restatement of the transformers-4.31 sampling loop,
    NewGenerationMixin.sample_stream (tortoise/models/stream_generator.py:722-1000, what api_fast.py:380-414 iterates), run on
    the reference's GPT2InferenceModel with the processors / warpers in 4.31's order (repetition penalty; temperature, top-k 50,
    top-p) and a 4.31-style boolean length criterion.  Same generator state =>
      * the codes equal oracle.ar_sample_loop bit for bit (ragged stop rows, whole-batch early exit, both position rules);
      * the per-step latents it yields (`final_norm(hidden_states[-1][:, -1])`) equal ONE teacher-forced pass over the codes:
        oracle.ar_latents with the plain positions for kv_cache=False and with the cached decode's positions 0, 2, 3, ... for
        kv_cache=True - which is how tortoise_tts_amd.api_fast.tts_stream obtains them (stages.ArStage.latents)."""
    import torch.nn.functional as F
    from transformers import (LogitsProcessorList, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                              TopPLogitsWarper)
    from oracle import make_golden as G
    SG = ref_shims.import_stream_generator()
    cfg = small_ar()
    sd = G.sampling_state_dict(cfg, eos_boost)
    m = ref_shims.enable_generate(G.build_ref_ar(ref, cfg, sd, kv_cache))
    cond, text = G.ar_inputs(cfg)
    t = F.pad(text, (0, 1), value=m.stop_text_token)                                  # inference_speech's prefix (autoregressive.py:535-556)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([cond.unsqueeze(1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]
    ids = torch.full((G.SAMPLE_B, P + 1), 1, dtype=torch.long)
    ids[:, -1] = m.start_mel_token
    max_len = P + 1 + G.SAMPLE_N
    torch.manual_seed(G.SAMPLE_SEED)
    toks, lats = [], []
    for tk, lat in SG.NewGenerationMixin.sample_stream(
            m.inference_model, ids, logits_processor=LogitsProcessorList([RepetitionPenaltyLogitsProcessor(2.0)]),
            logits_warper=LogitsProcessorList([TemperatureLogitsWarper(0.8), TopKLogitsWarper(50), TopPLogitsWarper(0.8)]),
            stopping_criteria=lambda input_ids, scores: input_ids.shape[-1] >= max_len,  # MaxLengthCriteria as 4.31 evaluated it
            pad_token_id=m.stop_mel_token, eos_token_id=[m.stop_mel_token], output_hidden_states=True, use_cache=True,
            attention_mask=torch.ones_like(ids)):
        toks.append(tk)
        lats.append(lat)
    codes, streamed = torch.stack(toks, 1), torch.stack(lats, 1)
    got = O.ar_sample_loop(sd, cfg, cond, text, G.SAMPLE_B, G.SAMPLE_N, G.sampling_noise(cfg), kv_cache=kv_cache)
    assert got.shape == codes.shape and torch.equal(got, codes)
    B = G.SAMPLE_B
    one_pass = O.ar_latents(sd, cfg, cond.expand(B, -1), text.expand(B, -1), codes, stream_positions=kv_cache)
    assert one_pass.shape == streamed.shape
    assert (one_pass - streamed).abs().max() < 2e-5
    if kv_cache:  # and the plain positions do NOT reproduce them under the cached rule (from step 1 on)
        plain = O.ar_latents(sd, cfg, cond.expand(B, -1), text.expand(B, -1), codes)
        assert (plain[:, 0] - streamed[:, 0]).abs().max() < 2e-5 and (plain[:, 1:] - streamed[:, 1:]).abs().max() > 1e-2


@torch.no_grad()
def test_ar_latents(ref):
    cfg = small_ar()
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=12)
    m = build_ref_ar(ref, cfg, sd)
    g = torch.Generator().manual_seed(1)
    k, n = 2, 24
    cond = torch.randn(1, cfg.model_dim, generator=g)
    text = torch.nn.functional.pad(torch.randint(1, 255, (1, 7), generator=g).int(), (0, 1))
    codes = torch.randint(0, 8192, (k, n), generator=g)
    want = m(cond.repeat(k, 1), text.repeat(k, 1), torch.tensor([text.shape[-1]]), codes.clone(),
             torch.tensor([n * m.mel_length_compression]), return_latent=True, clip_inputs=False)
    got = O.ar_latents(sd, cfg, cond.repeat(k, 1), text.repeat(k, 1), codes)
    assert got.shape == want.shape == (k, n, cfg.model_dim)
    assert torch.allclose(got, want, atol=2e-4, rtol=1e-4), (got - want).abs().max()


@torch.no_grad()
def test_clvp(ref):
    cfg = CLVPConfig(dim=128, dim_latent=128, depth=2, heads=2)
    sd = W.synthetic_state_dict(W.clvp_manifest(cfg), seed=13)
    m = ref.CLVP(dim_text=cfg.dim, dim_speech=cfg.dim, dim_latent=cfg.dim_latent, num_text_tokens=256,
                 text_enc_depth=cfg.depth, text_seq_len=350, text_heads=cfg.heads, num_speech_tokens=8192,
                 speech_enc_depth=cfg.depth, speech_heads=cfg.heads, speech_seq_len=430, use_xformers=True).eval()
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(2)
    text = torch.randint(0, 256, (1, 13), generator=g)
    codes = torch.randint(0, 8192, (3, 40), generator=g)
    want = m(text.repeat(3, 1), codes, return_loss=False)
    got = O.clvp_score(sd, cfg, text.repeat(3, 1), codes)
    assert torch.allclose(got, want, atol=1e-4, rtol=1e-4), (got - want).abs().max()


def build_ref_diffusion(ref, cfg, sd):
    m = ref.DiffusionTts(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
                         out_channels=cfg.out_channels, in_latent_channels=cfg.in_latent_channels,
                         in_tokens=cfg.in_tokens, dropout=0, use_fp16=False, num_heads=cfg.num_heads,
                         layer_drop=0, unconditioned_percentage=0).eval()
    m.load_state_dict(sd, strict=True)
    return m


@torch.no_grad()
def test_diffusion_network_and_sampler(ref):
    cfg = DiffusionConfig(model_channels=128, num_layers=2, in_latent_channels=128, num_heads=2)
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=14)
    m = build_ref_diffusion(ref, cfg, sd)
    g = torch.Generator().manual_seed(3)
    M = 12
    S = M * 4 * 24000 // 22050
    latents = torch.randn(1, M, cfg.in_latent_channels, generator=g)
    cond = torch.randn(1, 2 * cfg.model_channels, generator=g)
    want_emb = m.timestep_independent(latents, cond, S, False)
    got_emb = O.diffusion_timestep_independent(sd, cfg, latents, cond, S)
    assert torch.allclose(got_emb, want_emb, atol=1e-4, rtol=1e-4)
    x = torch.randn(1, 100, S, generator=g)
    ts = torch.tensor([2999])
    for cf in (False, True):
        want = m(x, ts, precomputed_aligned_embeddings=want_emb, conditioning_free=cf)
        got = O.diffusion_forward(sd, cfg, x, ts, got_emb, cf)
        assert torch.allclose(got, want, atol=2e-4, rtol=1e-4), (got - want).abs().max()
    # full sampler with injected noise: run the reference loop with a patched randn_like
    N = 5
    diffuser = ref.SpacedDiffusion(use_timesteps=ref.space_timesteps(4000, [N]), model_mean_type='epsilon',
                                   model_var_type='learned_range', loss_type='mse',
                                   betas=ref.get_named_beta_schedule('linear', 4000),
                                   conditioning_free=True, conditioning_free_k=2.0)
    sched = O.Schedule(N, 4000, True, 2.0)
    assert list(diffuser.timestep_map) == list(sched.timestep_map)
    step_noise = torch.randn(N, 1, 100, S, generator=g)
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
        want = diffuser.p_sample_loop(m, (1, 100, S), noise=x.clone(),
                                      model_kwargs={'precomputed_aligned_embeddings': want_emb}, progress=False)
    finally:
        rd.th.randn_like = orig
    got = O.p_sample_loop(sd, cfg, sched, got_emb, x.clone(), step_noise)
    assert torch.allclose(got, want, atol=5e-4, rtol=1e-3), (got - want).abs().max()


@torch.no_grad()
def test_conditioning_encoders(ref):
    """UnifiedVoice.get_conditioning / DiffusionTts.get_conditioning at the reference width of one block stack
    (autoregressive.py:204-228, 444-452; diffusion_decoder.py:186-192, 222-230)."""
    from oracle import make_golden as G
    a_cfg = ARConfig(layers=1, model_dim=256, heads=4)
    a_sd = W.synthetic_state_dict(W.ar_manifest(a_cfg), seed=31)
    m = G.build_ref_ar(ref, a_cfg, a_sd)
    g = torch.Generator().manual_seed(8)
    mel = torch.randn(1, 3, 80, 50, generator=g)
    want = m.get_conditioning(mel)
    got = O.ar_get_conditioning(a_sd, a_cfg, mel)
    assert got.shape == want.shape == (1, 256)
    assert torch.allclose(got, want, atol=1e-5), (got - want).abs().max()
    d_cfg = DiffusionConfig(model_channels=128, num_layers=1, in_latent_channels=128, num_heads=2)
    d_sd = W.synthetic_state_dict(W.diffusion_manifest(d_cfg), seed=32)
    d = ref.DiffusionTts(model_channels=d_cfg.model_channels, num_layers=d_cfg.num_layers, in_channels=d_cfg.in_channels,
                         out_channels=d_cfg.out_channels, in_latent_channels=d_cfg.in_latent_channels, in_tokens=d_cfg.in_tokens,
                         dropout=0, use_fp16=False, num_heads=d_cfg.num_heads, layer_drop=0, unconditioned_percentage=0).eval()
    d.load_state_dict(d_sd, strict=True)
    mel = torch.randn(1, 2, 100, 61, generator=g)  # odd length: the two stride-2 convolutions round up
    want = d.get_conditioning(mel)
    got = O.diffusion_get_conditioning(d_sd, d_cfg, mel)
    assert got.shape == want.shape == (1, 256)
    assert torch.allclose(got, want, atol=1e-5), (got - want).abs().max()


@torch.no_grad()
def test_univnet(ref):
    cfg = VocoderConfig()
    raw = W.synthetic_state_dict(W.vocoder_manifest(cfg), seed=15)
    m = ref.UnivNetGenerator()
    m.load_state_dict(raw, strict=True)
    m.eval(inference=True)
    sd = W.fold_weight_norm(raw)
    for k, v in m.state_dict().items():
        assert torch.allclose(sd[k], v, atol=1e-6), k
    g = torch.Generator().manual_seed(4)
    mel = torch.randn(1, 100, 6, generator=g) * 2 - 5
    z = torch.randn(1, 64, 16, generator=g)
    want = m.inference(mel, z)
    got = O.univnet_inference(sd, cfg, mel, z)
    assert got.shape == want.shape == (1, 1, 6 * 256)
    assert torch.allclose(got, want, atol=1e-4), (got - want).abs().max()
    assert got.abs().max() < 0.999 and got.abs().mean() > 1e-3  # not saturated, not dead


def test_integer_postprocessing_matches_reference_source(ref):
    """fix_autoregressive_output lives in tortoise/api.py, which cannot be imported here
    (progressbar/torchaudio/librosa missing), so its source is exec'd in isolation."""
    import ast, os, textwrap
    src = open(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "api.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fix_autoregressive_output"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "api_excerpt", "exec"), ns)
    rng = np.random.default_rng(0)
    for trial in range(50):
        n = int(rng.integers(4, 40))
        codes = rng.integers(0, 8192, n)
        if trial % 3:
            codes[int(rng.integers(0, n)):] = 8193
        if trial % 7 == 0 and n > 6:
            codes[int(rng.integers(0, n))] = 8193
        want = ns["fix_autoregressive_output"](torch.tensor(codes).clone(), 8193, complain=False).numpy()
        got = O.fix_autoregressive_output(codes, 8193)
        assert np.array_equal(want, got)


@pytest.mark.parametrize("c0,factors,kernels", [(64, [8, 8, 2, 2], [3, 7, 11]), (128, [4, 2], [3, 5]), (32, [2, 2, 2], [7])])
@torch.no_grad()
def test_hifigan_decoder(ref, c0, factors, kernels):
    """oracle.hifigan_inference vs the live HifiganGenerator.inference (hifigan_decoder.py:259-289) over several generator shapes
    (the streaming path's decoder)."""
    from tortoise.models.hifigan_decoder import HifiganGenerator
    from tortoise_tts_amd.config import HifiganConfig
    cfg = HifiganConfig(in_channels=48, cond_channels=40, upsample_initial_channel=c0, upsample_factors=factors,
                        upsample_kernel_sizes=[2 * u for u in factors], resblock_kernel_sizes=kernels)
    sd = W.synthetic_state_dict(W.hifigan_manifest(cfg), seed=c0)
    m = HifiganGenerator(in_channels=cfg.in_channels, out_channels=1, resblock_type="1",
                         resblock_dilation_sizes=[list(cfg.resblock_dilation_sizes)] * len(kernels), resblock_kernel_sizes=list(kernels),
                         upsample_kernel_sizes=list(cfg.upsample_kernel_sizes), upsample_initial_channel=c0,
                         upsample_factors=list(factors), cond_channels=cfg.cond_channels).eval()
    m.load_state_dict(sd, strict=True)
    m.device = torch.device("cpu")
    g = torch.Generator().manual_seed(1)
    lat, cond = torch.randn(1, 9, cfg.in_channels, generator=g), torch.randn(1, cfg.cond_channels, generator=g)
    want = m.inference(lat, cond)
    got = O.hifigan_inference(W.fold_weight_norm(sd), cfg, lat, cond)
    assert got.shape == want.shape
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-4), float((got - want).abs().max())
