"""Host logic of the drop-in class, driven on CPU with the exact call sequence of the reference's CLI
(tortoise/do_tts.py:31-47) and oracle-backed stand-ins for the GPU stages (tests/fake_stages.py).  What is checked is
everything api.py does BETWEEN the stage calls - flags, tokenisation, padding, fix_autoregressive_output, CLVP top-k,
calm-token trim, the return conventions of api.py:589-595 - plus the reference's error behaviour."""
import os

import pytest
import torch

from oracle import make_golden as G
from oracle import ref_shims
from tests import fake_stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig

VOCAB = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "data", "tokenizer.json")
PAT = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "voices", "cond_latent_example", "pat.pth")
DEFAULT_TEXT = "The expressiveness of autoregressive transformers is literally nuts! I absolutely adore them."  # do_tts.py:12


def small_setup():
    ar, clvp, diff = ARConfig(**G.AR_CFG), CLVPConfig(**G.CLVP_CFG), DiffusionConfig(**G.DIFF_CFG)
    sds = {"autoregressive": G.sampling_state_dict(ar, 2.0),  # stop token reachable: ragged candidates
           "clvp": W.synthetic_state_dict(W.clvp_manifest(clvp), seed=G.CLVP_SEED),
           "diffusion": W.synthetic_state_dict(W.diffusion_manifest(diff), seed=G.DIFF_SEED),
           "vocoder": W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(VocoderConfig()), seed=G.VOC_SEED)),
           "rlg_auto": W.synthetic_state_dict(W.rlg_manifest(ar.model_dim), seed=G.RLG_SEED, gain=3.0),
           "rlg_diffuser": W.synthetic_state_dict(W.rlg_manifest(2 * diff.model_channels), seed=G.RLG_SEED + 1, gain=3.0)}
    return sds, {"ar": ar, "clvp": clvp, "diffusion": diff}


def voice_latents(cfgs):
    """do_tts.py:39 `load_voices` on a .pth voice returns (None, latents) (utils/audio.py:104-124): the reference's own
    example file when the tree is here (1024 / 2048 wide), cut to the small test widths."""
    D, C2 = cfgs["ar"].model_dim, 2 * cfgs["diffusion"].model_channels
    if os.path.exists(PAT):
        a, d = torch.load(PAT, map_location="cpu")
        return a[:, :D].contiguous(), d[:, :C2].contiguous()
    g = torch.Generator().manual_seed(9)
    return torch.randn(1, D, generator=g) * 0.5, torch.randn(1, C2, generator=g) * 0.5


@pytest.fixture()
def tts(monkeypatch):
    if not os.path.exists(VOCAB):
        pytest.skip("tokenizer.json (reference data file) not present")
    fake_stages.install(monkeypatch)
    from tortoise_tts_amd.api import TextToSpeech
    sds, cfgs = small_setup()
    # do_tts.py:31: TextToSpeech(models_dir=..., use_deepspeed=..., kv_cache=..., half=...)  (+ engine-only keywords for the test sizes)
    t = TextToSpeech(models_dir="/nonexistent", use_deepspeed=False, kv_cache=True, half=True, tokenizer_vocab_file=VOCAB,
                     tokenizer_basic=True, state_dicts=sds, configs=cfgs, max_candidates=32, max_mel_tokens=48)
    t._cfgs = cfgs
    return t


@torch.no_grad()
def test_do_tts_call_sequence(tts):
    from tortoise_tts_amd import engine as E
    assert tts.dtype == E.TT_F16 and tts.kv_cache is True and tts.enable_redaction is True  # flags honoured, not ignored
    voice_samples, conditioning_latents = None, voice_latents(tts._cfgs)
    kw = dict(k=3, voice_samples=voice_samples, conditioning_latents=conditioning_latents, preset="ultra_fast",
              use_deterministic_seed=11, return_deterministic_state=True, cvvp_amount=0.0)
    gen, dbg = tts.tts_with_preset(DEFAULT_TEXT, max_mel_tokens=48, **kw)  # do_tts.py:41-42 (+ a short decode for CPU time)
    assert isinstance(gen, list) and len(gen) == 3  # api.py:589-592: k > 1 -> list of k clips
    for g_ in gen:
        assert g_.dim() == 3 and g_.shape[:2] == (1, 1) and g_.dtype == torch.float32 and g_.device.type == "cpu"
        assert g_.shape[-1] % 256 == 0 and g_.shape[-1] > 0 and torch.isfinite(g_).all() and g_.abs().max() <= 1.0
        assert g_.squeeze(0).cpu().shape[0] == 1  # what do_tts.py:45 hands torchaudio.save
    seed, text, vs, lat = dbg  # api.py:594-595
    assert seed == 11 and text == DEFAULT_TEXT and vs is None and lat is conditioning_latents
    # the ranked winners are fix_autoregressive_output'ed rows padded to max_mel_tokens (api.py:425-426, 459)
    best = tts.last_best_codes
    assert best.shape == (3, 48) and best.max() < 8193
    # same seed -> same audio; k = 1 returns a bare tensor (api.py:591-592)
    again = tts.tts_with_preset(DEFAULT_TEXT, max_mel_tokens=48, **dict(kw, k=1, return_deterministic_state=False))
    assert torch.is_tensor(again) and torch.equal(again, gen[0])


@torch.no_grad()
def test_tts_many_equals_one_utterance_after_the_other(monkeypatch):
    """tts_many (the long-form path: read.py:66-71 renders its chunks one after the other with the same seed) batches the
    autoregressive stage over `utterance_batch` utterances; everything it returns must equal tts() called per text."""
    if not os.path.exists(VOCAB):
        pytest.skip("tokenizer.json (reference data file) not present")
    fake_stages.install(monkeypatch)
    from tortoise_tts_amd.api import TextToSpeech
    sds, cfgs = small_setup()
    t = TextToSpeech(models_dir="/nonexistent", tokenizer_vocab_file=VOCAB, tokenizer_basic=True, state_dicts=sds, configs=cfgs,
                     max_candidates=8, max_mel_tokens=40, candidate_sharding=False, utterance_batch=2)
    lat = voice_latents(cfgs)
    texts = ["Once upon a time.", "There lived a girl.", list(range(10, 31))]
    kw = dict(num_autoregressive_samples=8, diffusion_iterations=3, max_mel_tokens=32)
    one_by_one = [t.tts(x, conditioning_latents=lat, use_deterministic_seed=5, verbose=False, **kw) for x in texts]
    many = t.tts_many(texts, conditioning_latents=lat, use_deterministic_seed=5, **kw)
    assert t.ar.group_batches == 1  # 3 utterances, 2 per decode batch: one grouped generation + one single
    assert t.diffusion.batched == [2]  # and the same for the denoiser passes: the two nearest in length together, one alone
    assert len(many) == 3 and all(torch.equal(a, b) for a, b in zip(many, one_by_one))
    assert set(t.timings) >= {"ar_s", "diffusion_s", "total_s"}
    # what the grouped decode cannot hold falls back to tts() per utterance (same results): a candidate count that is not a multiple of 4
    odd = t.tts_many(texts[:2], conditioning_latents=lat, use_deterministic_seed=5, num_autoregressive_samples=6, diffusion_iterations=3, max_mel_tokens=32)
    want = [t.tts(x, conditioning_latents=lat, use_deterministic_seed=5, verbose=False, num_autoregressive_samples=6, diffusion_iterations=3, max_mel_tokens=32)
            for x in texts[:2]]
    assert all(torch.equal(a, b) for a, b in zip(odd, want))
    with pytest.raises(NotImplementedError):
        t.tts_many(texts, conditioning_latents=lat, k=2, **kw)
    # typical sampling (tts(typical_sampling=True, typical_mass=...), api.py:361-364) rides through both entry points: the grouped decode
    # equals the per-utterance calls, and it is not a no-op
    typ = dict(kw, typical_sampling=True, typical_mass=0.3)
    one_typ = [t.tts(x, conditioning_latents=lat, use_deterministic_seed=5, verbose=False, **typ) for x in texts[:2]]
    codes_typ = t.last_best_codes.clone()
    many_typ = t.tts_many(texts[:2], conditioning_latents=lat, use_deterministic_seed=5, **typ)
    assert all(torch.equal(a, b) for a, b in zip(many_typ, one_typ))
    t.tts(texts[1], conditioning_latents=lat, use_deterministic_seed=5, verbose=False, **kw)
    assert not torch.equal(t.last_best_codes, codes_typ)
    with pytest.raises(NotImplementedError, match="num_beams"):
        t.tts(texts[0], conditioning_latents=lat, num_beams=2, **kw)
    with pytest.raises(ValueError, match="typical_mass"):
        t.tts(texts[0], conditioning_latents=lat, typical_sampling=True, typical_mass=0.0, **kw)


@torch.no_grad()
def test_random_voice_and_error_behaviour(tts):
    # voice='random' (do_tts.py default): no samples, no latents -> RandomLatentConverter pair (api.py:398-399, 301-309)
    torch.manual_seed(3)
    a, d = tts.get_random_conditioning_latents()
    assert a.shape == (1, tts._cfgs["ar"].model_dim) and d.shape == (1, 2 * tts._cfgs["diffusion"].model_channels)
    torch.manual_seed(3)
    a2, _ = tts.get_random_conditioning_latents()
    assert torch.equal(a, a2)
    wav = tts.tts("hello there", num_autoregressive_samples=4, diffusion_iterations=4, max_mel_tokens=24, use_deterministic_seed=1)
    assert torch.is_tensor(wav) and wav.shape[:2] == (1, 1)
    with pytest.raises(ValueError, match="Too much text"):  # api.py:392
        tts.tts(list(range(1, 255)) * 2, conditioning_latents=(a, d))
    with pytest.raises(NotImplementedError, match="bracket"):
        tts.tts("[I am so sad,] hello", conditioning_latents=(a, d))
    with pytest.raises(ValueError, match="cvvp_amount"):
        tts.tts("hello", conditioning_latents=(a, d), cvvp_amount=1.5)
    with pytest.raises(ValueError, match="max_mel_tokens"):
        tts.tts("hello", conditioning_latents=(a, d), max_mel_tokens=500)


@torch.no_grad()
def test_voice_samples_path(tts):
    """do_tts.py with a wav voice: load_voices returns (clips, None) and tts() calls get_conditioning_latents(voice_samples)
    (api.py:396-397, 258-299).  Raw 22.05 kHz clips go through the torch mel front-end (audio.py), ready mel pairs are taken as is."""
    from tortoise_tts_amd.audio import MelFrontEnd
    g = torch.Generator().manual_seed(5)
    clips = [torch.randn(1, 30000, generator=g).clamp(-1, 1) * 0.2, torch.randn(1, 45000, generator=g).clamp(-1, 1) * 0.2]
    tts.mel_front_end = MelFrontEnd(mel_norms=torch.ones(80))  # data/mel_norms.pth may be absent on the test box
    a, d, am, dm = tts.get_conditioning_latents(clips, return_mels=True)
    assert a.shape == (1, tts._cfgs["ar"].model_dim) and d.shape == (1, 2 * tts._cfgs["diffusion"].model_channels)
    assert am.shape[:3] == (1, 2, 80) and dm.shape[:3] == (1, 2, 100)  # api.py:275, 288: clips stacked on dim 1
    pairs = [(am[:, 0], dm[:, 0]), (am[:, 1], dm[:, 1])]
    a2, d2 = tts.get_conditioning_latents(pairs)
    assert torch.equal(a, a2) and torch.equal(d, d2)
    wav = tts.tts("hello there", voice_samples=clips, num_autoregressive_samples=4, diffusion_iterations=4, max_mel_tokens=24,
                  use_deterministic_seed=2)
    assert torch.is_tensor(wav) and wav.shape[:2] == (1, 1) and torch.isfinite(wav).all()


@torch.no_grad()
def test_cvvp_amount_blends_the_candidate_ranking(tts):
    """tts(cvvp_amount > 0) (api.py:450-472): the CVVP model is built on first use, scores the candidates against the voice's conditioning
    clips (mean over the clips) and is blended with CLVP as cvvp * amount + clvp * (1 - amount); amount == 1 ranks by CVVP alone; without
    clips (latents-only voice) the ranking stays CLVP's, as upstream."""
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd import weights as W
    from tortoise_tts_amd.config import CVVPConfig
    ccfg = CVVPConfig(**G.CVVP_CFG)
    tts.cvvp_cfg = ccfg
    tts._state_dicts["cvvp"] = W.synthetic_state_dict(W.cvvp_manifest(ccfg), seed=G.CVVP_SEED)
    g = torch.Generator().manual_seed(8)
    pairs = [(torch.randn(1, 80, 64, generator=g) * 2 - 5, torch.randn(1, 100, 70, generator=g) * 2 - 5) for _ in range(2)]
    kw = dict(voice_samples=pairs, num_autoregressive_samples=8, diffusion_iterations=3, max_mel_tokens=24, use_deterministic_seed=3, k=2)
    assert tts.cvvp is None
    tts.tts("hello there", **kw)
    assert tts.cvvp is None  # "only loaded if used" (api.py:234)
    clvp_best = tts.last_best_codes.clone()
    tts.tts("hello there", cvvp_amount=1.0, **kw)
    assert tts.cvvp is not None and tts.cvvp.calls == 1
    cvvp_best = tts.last_best_codes.clone()
    # recompute both rankings from the candidates (same seed -> same candidates; the winners are rows of the same candidate set)
    auto_conds = torch.stack([p[0] for p in pairs], dim=1)
    cand = tts.ar.gen_codes
    from tortoise_tts_amd.api import fix_autoregressive_output
    import torch.nn.functional as F
    fixed = fix_autoregressive_output(F.pad(cand, (0, 24 - cand.shape[1]), value=tts.stop_mel_token), tts.stop_mel_token)
    want = fixed[torch.topk(O.cvvp_score(tts._state_dicts["cvvp"], ccfg, auto_conds, fixed), 2).indices]
    assert torch.equal(cvvp_best, want) and not torch.equal(cvvp_best, clvp_best)
    tts.tts("hello there", cvvp_amount=0.5, **kw)
    text = torch.as_tensor(tts.tokenizer.encode("hello there")).reshape(1, -1)
    clvp = O.clvp_score(tts._state_dicts["clvp"], tts._cfgs["clvp"], F.pad(text, (0, 1)).repeat(8, 1), fixed)
    blend = O.blend_candidate_scores(clvp, O.cvvp_score(tts._state_dicts["cvvp"], ccfg, auto_conds, fixed), 0.5)
    assert torch.equal(tts.last_best_codes, fixed[torch.topk(blend, 2).indices])
    # latents only: nothing for CVVP to compare with -> CLVP's ranking (api.py:464, 473); amount 1 has nothing to rank by at all
    lat = voice_latents(tts._cfgs)
    kw2 = dict(kw, voice_samples=None, conditioning_latents=lat)
    tts.tts("hello there", **kw2)
    only_clvp = tts.last_best_codes.clone()
    n_calls = tts.cvvp.calls
    tts.tts("hello there", cvvp_amount=0.5, **kw2)
    assert torch.equal(tts.last_best_codes, only_clvp) and tts.cvvp.calls == n_calls
    with pytest.raises(ValueError, match="voice_samples"):
        tts.tts("hello there", cvvp_amount=1.0, **kw2)


def test_constructor_flags(monkeypatch):
    fake_stages.install(monkeypatch)
    from tortoise_tts_amd.api import TextToSpeech
    from tortoise_tts_amd import engine as E
    sds, cfgs = small_setup()
    kw = dict(state_dicts=sds, configs=cfgs, max_candidates=8, max_mel_tokens=16)
    t = TextToSpeech(**kw)  # reference defaults: kv_cache=False, half=False
    assert t.kv_cache is False and t.ar.kv_cache is False and t.dtype == E.TT_F16  # round 6: fp16 + overflow guard is the default operand type
    assert TextToSpeech(dtype="bf16", **kw).dtype == E.TT_BF16
    assert TextToSpeech(half=True, **kw).dtype == E.TT_F16
    assert TextToSpeech(dtype="fp16", **kw).dtype == E.TT_F16
    with pytest.raises(ValueError):
        TextToSpeech(half=True, dtype="bf16", **kw)
    with pytest.raises(NotImplementedError):
        TextToSpeech(use_deepspeed=True, **kw)
    assert TextToSpeech(autoregressive_batch_size=64, **kw).autoregressive_batch_size == 8  # clamped to the handle's capacity
