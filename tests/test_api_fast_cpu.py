"""CPU checks of the streaming surface (tortoise_tts_amd/api_fast.py): handle_chunks against the reference's own method
(tortoise/api_fast.py:275-309, extracted from the source file because the module itself does not import offline), and the
class keeps the reference's signatures."""
import ast
import inspect
import os

import pytest
import torch

from oracle import ref_shims

REF = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "api_fast.py")


def reference_method(name):
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TextToSpeech"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    ns = {"torch": torch, "MODELS_DIR": None}  # (a default-argument name of __init__)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "api_fast_excerpt", "exec"), ns)
    return fn, ns[name]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_handle_chunks_matches_the_reference_method():
    from tortoise_tts_amd.api_fast import TextToSpeech
    _, ref_fn = reference_method("handle_chunks")
    g = torch.Generator().manual_seed(0)
    for overlap in (64, 256):
        prev_r = over_r = prev_m = over_m = None
        length = 0
        for grow in (900, 300, 40, 700, 10):  # growing decodes of the latents so far; one piece shorter than the overlap
            length += grow
            wav = torch.randn(length, generator=g)
            cr, prev_r, over_r = ref_fn(None, wav.clone(), prev_r, over_r, overlap)
            cm, prev_m, over_m = TextToSpeech.handle_chunks(wav.clone(), prev_m, over_m, overlap)
            assert torch.equal(cr, cm)
            assert (over_r is None) == (over_m is None) and (over_r is None or torch.equal(over_r, over_m))
            assert prev_r.shape == prev_m.shape  # only its length is used by the next call (the reference cross-fades it in place)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_signatures_follow_the_reference():
    from tortoise_tts_amd.api_fast import TextToSpeech
    for name in ("tts_stream", "tts"):
        fn, _ = reference_method(name)
        ref_args = [a.arg for a in fn.args.args][1:]
        ours = list(inspect.signature(getattr(TextToSpeech, name)).parameters)[1:]
        for a in ref_args:
            assert a in ours, f"{name}: reference parameter {a!r} missing"
    fn, _ = reference_method("__init__")
    ref_args = [a.arg for a in fn.args.args][1:]
    ours = list(inspect.signature(TextToSpeech.__init__).parameters)[1:]
    assert ours[:len(ref_args)] == ref_args


@torch.no_grad()
def test_tts_and_stream_flow_on_oracle_backed_stages(monkeypatch):
    """api_fast host logic end to end on CPU stand-ins (tests/fake_stages.py): tts() = 1 sampled sequence -> teacher-forced latents ->
    HiFi-GAN; tts_stream() emits the first piece after 60 tokens, then every stream_chunk_size tokens, cross-faded; the pieces add up
    to the one-shot waveform minus the last overlap window."""
    from oracle import make_golden as G
    from tests import fake_stages
    from tortoise_tts_amd import weights as W
    from tortoise_tts_amd.config import ARConfig, HifiganConfig
    fake_stages.install(monkeypatch)
    from tortoise_tts_amd import api_fast
    monkeypatch.setattr(api_fast.E, "require_gpu", lambda device=None: torch.device("cpu"))
    a_cfg = ARConfig(**G.AR_CFG)
    h_cfg = HifiganConfig(in_channels=a_cfg.model_dim, cond_channels=a_cfg.model_dim, upsample_initial_channel=64)
    sds = {"autoregressive": W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(a_cfg), seed=G.AR_SEED), a_cfg),
           "hifidecoder": W.synthetic_state_dict(W.hifigan_manifest(h_cfg), seed=43),
           "rlg_auto": W.synthetic_state_dict(W.rlg_manifest(a_cfg.model_dim), seed=G.RLG_SEED, gain=3.0)}
    tts = api_fast.TextToSpeech(state_dicts=sds, configs={"ar": a_cfg, "hifigan": h_cfg}, max_mel_tokens=80, max_text_tokens=40, kv_cache=True)
    text = list(range(5, 20))
    wav = tts.tts(text, max_mel_tokens=70, use_deterministic_seed=4)  # random voice (api_fast.py:375)
    assert wav.dim() == 3 and wav.shape[:2] == (1, 1) and torch.isfinite(wav).all() and wav.abs().max() <= 1.0
    torch.manual_seed(0)
    chunks = list(tts.tts_stream(text, max_mel_tokens=70, use_deterministic_seed=4, stream_chunk_size=5, overlap_wav_len=128))
    # 60 tokens (first buffer), 65, 70 - and, because the 70-token limit falls exactly on a buffer boundary, the reference's final
    # StopIteration pass over the same latents, which hands out the withheld overlap window (api_fast.py:405-420, 286-291)
    assert len(chunks) == 4 and chunks[-1].shape[0] == 128
    assert sum(int(c.shape[0]) for c in chunks) == wav.shape[-1]
    assert torch.equal(tts.last_codes, tts.last_codes[:, :70])
    # latents: tts() re-passes with the plain positions (api_fast.py:510-514); the stream takes the latents its decode steps filed
    assert tts.ar.latent_calls == [False, "steps", "steps", "steps"]
    # the earlier formulation (one teacher-forced pass per chunk) asks for the cached decode's positions because this instance was
    # built with kv_cache=True; with kv_cache=False it is the plain pass.  Same audio either way (oracle stand-ins: identical tensors)
    tts.stream_latents_from = "pass"
    tts.ar.latent_calls = []
    again = list(tts.tts_stream(text, max_mel_tokens=70, use_deterministic_seed=4, stream_chunk_size=5, overlap_wav_len=128))
    assert tts.ar.latent_calls == [True, True, True]
    assert len(again) == len(chunks) and all(torch.equal(a, b) for a, b in zip(again, chunks))
    tts2 = api_fast.TextToSpeech(state_dicts=sds, configs={"ar": a_cfg, "hifigan": h_cfg}, max_mel_tokens=80, max_text_tokens=40)
    tts2.stream_latents_from = "pass"
    list(tts2.tts_stream(text, max_mel_tokens=64, use_deterministic_seed=4, stream_chunk_size=5, overlap_wav_len=128))
    assert tts2.ar.latent_calls == [False, False]
    with pytest.raises(ValueError, match="Too much text"):
        tts.tts(list(range(1, 255)) * 2)
    tts.tts(text, max_mel_tokens=40, use_deterministic_seed=4)
    base = tts.last_codes.clone()
    tts.tts(text, max_mel_tokens=40, use_deterministic_seed=4, cvvp_amount=0.5)  # accepted and unused, as upstream (api_fast.py:426: one sample, no ranking)
    assert torch.equal(tts.last_codes, base)
    with pytest.raises(NotImplementedError, match="num_beams"):
        list(tts.tts_stream(text, num_beams=4))
    # typical sampling (api.py:361-364 -> autoregressive.py:558) is honoured: other codes than plain sampling on the same seed, the
    # mass only counts with the switch on (as upstream), an impossible mass is refused
    tts.tts(text, max_mel_tokens=40, use_deterministic_seed=4)
    plain = tts.last_codes.clone()
    tts.tts(text, max_mel_tokens=40, use_deterministic_seed=4, typical_mass=0.2)
    assert torch.equal(tts.last_codes, plain)
    tts.tts(text, max_mel_tokens=40, use_deterministic_seed=4, typical_sampling=True, typical_mass=0.2)
    assert tts.last_codes.shape != plain.shape or not torch.equal(tts.last_codes, plain)
    with pytest.raises(ValueError, match="typical_mass"):
        tts.tts(text, typical_sampling=True, typical_mass=1.0)


def reference_decode_points(n_pairs, stream_chunk_size):
    """The reference's streaming loop (api_fast.py:399-420) restated over a generator that yields n_pairs (token, latent) pairs:
    the number of latents handed to hifi_decoder.inference at every decode, in order."""
    gen = iter(range(n_pairs))
    points, all_latents, codes_, is_end, first_buffer = [], [], [], False, 60
    while not is_end:
        try:
            all_latents.append(next(gen))
            codes_.append(0)
        except StopIteration:
            is_end = True
        if is_end or (stream_chunk_size > 0 and len(codes_) >= max(stream_chunk_size, first_buffer)):
            first_buffer = 0
            points.append(len(all_latents))
            codes_ = []
    return points


@pytest.mark.parametrize("n_tokens,eos,chunk", [(70, False, 5), (64, False, 5), (61, True, 40), (60, True, 40), (100, True, 40), (99, False, 40),
                                                (140, False, 40), (30, True, 40), (59, False, 20), (120, False, 20)])
@torch.no_grad()
def test_stream_decode_points_follow_the_reference_loop(monkeypatch, n_tokens, eos, chunk):
    """Which prefixes of the latent sequence get decoded, and how often: the engine-side loop (resumable chunks + `done`) against
    the reference's pull loop, including the extra decode of an unchanged prefix when the sequence ends on a buffer boundary.
    eos: the sequence ends with a sampled stop token (which the reference's generator never yields) instead of the length limit."""
    from tortoise_tts_amd import api_fast
    stop = 8193

    class Ar:
        def prefill(self, *a):
            pass

        def generate_stream(self, B, max_new, chunk, first_chunk=None, **kw):
            codes = torch.arange(n_tokens + (1 if eos else 0))[None] % 50
            if eos:
                codes[0, -1] = stop
            total, pos, first = codes.shape[1], 0, True
            while pos < total:
                pos = min(pos + ((first_chunk or chunk) if first else chunk), total)
                first = False
                yield codes[:, :pos], pos >= total

        def latents(self, cond, text, codes, stream_positions=False):
            return torch.zeros(1, codes.shape[1], 4)

    seen = []

    class Hifi:
        def inference(self, latents, g):
            seen.append(latents.shape[1])
            return torch.arange(latents.shape[1] * 256, dtype=torch.float32)[None, None]

    tts = api_fast.TextToSpeech.__new__(api_fast.TextToSpeech)
    tts.ar, tts.hifi_decoder, tts.kv_cache, tts.stop_mel_token, tts.max_mel_tokens_cap = Ar(), Hifi(), False, stop, 500
    tts.stream_latents_from = "pass"
    tts.device = torch.device("cpu")
    monkeypatch.setattr(api_fast.TextToSpeech, "_prepare", lambda self, *a: (torch.zeros(1, 4, dtype=torch.int32), torch.zeros(1, 4)))
    limit = n_tokens if not eos else 500
    pieces = list(tts.tts_stream([1, 2, 3], max_mel_tokens=limit, stream_chunk_size=chunk, overlap_wav_len=64, use_deterministic_seed=1))
    want = reference_decode_points(n_tokens, chunk)
    # the engine decodes an unchanged prefix only once and re-runs handle_chunks on it: compare piece count and decoded prefixes
    assert len(pieces) == len(want)
    assert sorted(set(want)) == sorted(set(seen))
    # and the pieces are what the reference's handle_chunks makes of those decodes
    _, ref_fn = reference_method("handle_chunks")
    prev = over = None
    for n, got in zip(want, pieces):
        wav = torch.arange(n * 256, dtype=torch.float32)
        ch, prev, over = ref_fn(None, wav, prev, over, 64)
        assert torch.equal(ch, got)


@torch.no_grad()
def test_latent_pass_embeddings_follow_the_oracle():
    """stages.latent_pass_embeddings (the host side of ArStage.latents, plain tensor indexing) builds the same input rows as the
    oracle's teacher-forced pass, for the plain positions and for the cached decode's 0, 2, 3, ... rule."""
    from oracle import make_golden as G
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd import stages
    from tortoise_tts_amd import weights as W
    from tortoise_tts_amd.config import ARConfig
    cfg = ARConfig(**G.AR_CFG)
    sd = W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED)
    cond, text = G.ar_inputs(cfg)
    g = torch.Generator().manual_seed(2)
    codes = torch.randint(0, cfg.number_mel_codes - 2, (3, 17), generator=g)
    for flag in (False, True):
        emb, mel_rows = stages.latent_pass_embeddings(sd["text_embedding.weight"], sd["text_pos_embedding.emb.weight"], sd["mel_embedding.weight"],
                                                      sd["mel_pos_embedding.emb.weight"], cfg, cond, text, codes, flag)
        assert mel_rows == 17 + 2 and emb.shape == (3, 1 + text.shape[1] + 2 + 19, cfg.model_dim)
        hidden, _ = O.gpt2_trunk(sd, cfg, emb)
        enc = torch.nn.functional.layer_norm(hidden[:, 1:], (cfg.model_dim,), sd["final_norm.weight"], sd["final_norm.bias"], 1e-5)
        got = enc[:, -mel_rows:][:, :-2]
        want = O.ar_latents(sd, cfg, cond.expand(3, -1), text.expand(3, -1), codes, stream_positions=flag)
        assert torch.equal(got, want)
    plain, _ = stages.latent_pass_embeddings(sd["text_embedding.weight"], sd["text_pos_embedding.emb.weight"], sd["mel_embedding.weight"],
                                             sd["mel_pos_embedding.emb.weight"], cfg, cond, text, codes, False)
    assert torch.equal(plain[:, :-18], emb[:, :-18]) and not torch.equal(plain[:, -18:], emb[:, -18:])  # only mel inputs 1.. move
