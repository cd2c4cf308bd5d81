"""-m gpu: BASELINE-size PROPERTY checks (reference hyper-parameters: 30x1024 GPT-2, 10-layer 1024-wide DiffusionTts,
20-layer CLVP towers, UnivNet).  Numerical parity at these sizes against the reference modules and the oracle lives in
tests/test_gpu_fullsize.py; this file adds the size-independent properties:
  * KV-cached decode == teacher-forced full pass fed the same position rows (two different kernel paths);
  * sampled codes: in range, never the suppressed stop token, bit-reproducible, invariant to candidate sharding;
  * batched cond/uncond denoiser row == stand-alone conditioned evaluation; hipGraph replay == eager launches;
  * vocoder / end-to-end output shape (S * 256 samples), finiteness, clamp range, determinism per seed."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import report, rel_err
from tortoise_tts_amd import engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    import bench
    from tortoise_tts_amd.api import TextToSpeech
    sds = bench.synthetic_weights()
    text, latents = bench.synthetic_prompt()
    tts = TextToSpeech(state_dicts=sds, dtype="bf16", max_candidates=32, max_mel_tokens=200, kv_cache=True)
    return tts, sds, text, latents


@torch.no_grad()
def test_cached_decode_equals_full_pass(full):
    tts, sds, text, latents = full
    ar, cfg = tts.ar, tts.ar_cfg
    dev = tts.device
    auto = latents[0].to(dev)
    tt = F.pad(text.int()[None].to(dev), (0, 1))
    B, steps = 4, 6
    g = torch.Generator().manual_seed(3)
    toks = torch.randint(0, 8192, (steps, B), generator=g)
    ar.prefill(auto, tt)
    got = [ar.logits(1).expand(B, -1).clone()]
    ar.begin(B)
    for s in range(steps):
        ar.decode_step(toks[s])
        got.append(ar.logits(B).clone())
    got = torch.stack(got)  # [steps+1, B, V]
    # full pass over [prefix | start | tok_0 .. tok_{steps-1}] with the KV-cache position rows 0, 2, 3, ...
    prefix = ar.prefix_embedding(auto, tt)  # [1, P, D]
    start = ar.w.mel_emb[cfg.start_mel_token] + ar.w.mel_pos[0]
    rows = [ar.w.mel_emb[toks[s].to(dev)] + ar.w.mel_pos[s + 2] for s in range(steps)]  # each [B, D]
    mel = torch.stack([start[None].expand(B, -1)] + rows, dim=1)  # [B, steps+1, D]
    emb = torch.cat([prefix.expand(B, -1, -1), mel], dim=1).contiguous()
    out = torch.empty_like(emb)
    from tortoise_tts_amd import engine as E
    E.check(ar.lib.tt_ar_latents(ar.h, E.ptr(emb), B, emb.shape[1], E.ptr(out), E.stream_ptr()))
    Wh = sds["autoregressive"]["mel_head.weight"].to(dev).bfloat16().float()
    bh = sds["autoregressive"]["mel_head.bias"].to(dev)
    h = out[:, -(steps + 1):].bfloat16().float()  # the engine feeds the head GEMM operand-rounded activations
    full_logits = (h @ Wh.t() + bh).permute(1, 0, 2)
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool, device=dev)
    keep[cfg.stop_mel_token] = False  # suppressed (-1e9) logit would dominate the norm
    report("full-size AR: KV-cached decode vs teacher-forced full pass", got[..., keep], full_logits[..., keep], 2e-2)


@torch.no_grad()
def test_sampling_properties_full_size(full):
    tts, sds, text, latents = full
    ar, cfg = tts.ar, tts.ar_cfg
    auto = latents[0].to(tts.device)
    tt = F.pad(text.int()[None].to(tts.device), (0, 1))

    def gen(B, off):
        ar.prefill(auto, tt)
        return ar.generate(B, 24, seed=5, row_offset=off)[0]
    a = gen(32, 0)
    assert a.shape == (32, 24) and a.min() >= 0 and a.max() < cfg.number_mel_codes
    assert (a != cfg.stop_mel_token).all(), "suppressed stop token was sampled"
    assert torch.equal(a, gen(32, 0)), "same seed gave different codes"
    assert torch.equal(a, torch.cat([gen(16, 0), gen(16, 16)])), "sharding the candidates changed the codes"
    assert len(torch.unique(a[:, 0])) > 1, "candidates are not independent samples"


@torch.no_grad()
def test_diffusion_full_size_properties(full):
    tts, sds, text, latents = full
    from tortoise_tts_amd.schedule import Schedule
    df = tts.diffusion
    dev = tts.device
    g = torch.Generator().manual_seed(0)
    M = 200
    S = M * 4 * 24000 // 22050
    lat = torch.randn(1, M, 1024, generator=g)
    df.condition(lat, latents[1], S)
    x = torch.randn(1, 100, S, generator=g)
    both = df.forward(x, 1500, cond_free=True)
    one = df.forward(x, 1500, cond_free=False)
    assert torch.isfinite(both).all()
    assert rel_err(one[0], both[0]) < 1e-6, "conditioned row depends on batching"
    assert rel_err(both[1], both[0]) > 1e-3, "unconditioned row equals the conditioned one"
    sched = Schedule(30, 4000, True, 2.0)
    sched.num_timesteps = 4  # first 4 table rows only: keeps the test short
    sched.timestep_map = sched.timestep_map[:4]
    noise = torch.randn(4, 1, 100, S, generator=g)
    mel = df.sample(sched, x, noise)
    E.load_library().tt_graph_replay(0)
    try:
        mel2 = df.sample(sched, x, noise)
    finally:
        E.load_library().tt_graph_replay(1)
    assert torch.equal(mel, mel2) and torch.isfinite(mel).all()
    assert mel.shape == (1, 100, S) and mel.min() >= -11.6 and mel.max() <= 2.4  # x0 clamp -> tacotron range


@torch.no_grad()
def test_vocoder_and_end_to_end(full):
    tts, sds, text, latents = full
    g = torch.Generator().manual_seed(1)
    S = 870
    wav = tts.vocoder.inference(torch.randn(1, 100, S, generator=g) * 2 - 5, torch.randn(1, 64, S + 10, generator=g))
    assert wav.shape == (1, 1, S * 256) and torch.isfinite(wav).all() and wav.abs().max() <= 1.0 and wav.abs().mean() > 1e-4
    kw = dict(conditioning_latents=latents, num_autoregressive_samples=16, diffusion_iterations=6, max_mel_tokens=40,
              use_deterministic_seed=7, verbose=False)
    a = tts.tts(text, **kw)
    b = tts.tts(text, **kw)
    S = 40 * 4 * 24000 // 22050
    assert a.shape == (1, 1, S * 256) and a.device.type == "cpu" and torch.isfinite(a).all() and a.abs().max() <= 1.0
    assert torch.equal(a, b), "same deterministic seed gave different audio"
    c = tts.tts(text, **dict(kw, use_deterministic_seed=8))
    assert not torch.equal(a, c)
    print("[parity] end-to-end stage seconds:", {k: round(v, 4) for k, v in tts.timings.items()})


@torch.no_grad()
def test_tts_many_shares_decode_batches_and_equals_tts_per_text(full):
    """Long-form path: TextToSpeech(utterance_batch=G).tts_many decodes the candidates of G utterances in one batch (own prefix and
    Philox key per utterance); every clip must be bit-identical to tts() on that text alone - full-width engines, three texts of
    different lengths, two per decode batch."""
    import bench
    from tortoise_tts_amd.api import TextToSpeech
    _, sds, _, latents = full
    tts = TextToSpeech(state_dicts=sds, dtype="bf16", max_candidates=16, max_mel_tokens=64, kv_cache=True, candidate_sharding=False,
                       utterance_batch=2)
    g = torch.Generator().manual_seed(11)
    texts = [torch.randint(1, 255, (n,), generator=g) for n in (23, 61, 40)]
    kw = dict(conditioning_latents=latents, num_autoregressive_samples=16, diffusion_iterations=4, max_mel_tokens=36)
    one_by_one, codes = [], []
    for t in texts:
        one_by_one.append(tts.tts(t, use_deterministic_seed=7, verbose=False, **kw))
        codes.append(tts.last_best_codes.clone())
    tts.batch_diffusion = False  # shared decode batches only: every clip bit-identical to tts() alone
    many = tts.tts_many(texts, use_deterministic_seed=7, **kw)
    assert len(many) == 3
    for a, b in zip(many, one_by_one):
        assert a.shape == b.shape and torch.equal(a, b), "an utterance rendered inside a shared decode batch differs from tts() alone"
    tts.batch_diffusion = True   # + shared denoiser passes (padded to the longest): same winner, clip within the operand tolerance
    many = tts.tts_many(texts, use_deterministic_seed=7, **kw)
    assert torch.equal(tts.last_best_codes, codes[-1])
    for j, (a, b) in enumerate(zip(many, one_by_one)):
        assert a.shape == b.shape and torch.isfinite(a).all()
        report(f"tts_many (batched AR + batched denoiser) clip {j} bf16 vs tts() alone", a, b, 8e-2)
    again = tts.tts_many(texts, use_deterministic_seed=7, **kw)
    assert all(torch.equal(a, b) for a, b in zip(many, again)), "tts_many is not reproducible run to run"
    assert tts.timings["ar_s"] > 0
    for st in (tts.ar, tts.clvp, tts.diffusion, tts.vocoder):
        st.close()
