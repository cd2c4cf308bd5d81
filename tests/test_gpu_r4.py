"""-m gpu, round 4: what changed in the engines' control structure, each asserted as an EQUALITY (these are scheduling changes, the
arithmetic of a row must not move):
  * (the row-range option of round 4 lost its A/B and left the product in round 5: profiles/r04_ab_ar_subbatches.txt keeps the record;
    the repeated bit-identity checks it introduced now guard the decode step itself, tests/test_gpu_r5.py);
  * the launch loop paced by progress words in pinned memory (no queue drain inside the loop): same codes, same early exit;
  * seeds, row_offset and the caller's code buffer are DATA of the kept decode-step graph (one capture for all of them);
  * the sampler-step graph of the diffusion stage stays on the handle (one capture for several calls with fresh tensors);
  * the operand-overflow guards trip on fp16 overflow, stay silent otherwise, and TextToSpeech re-renders with bf16 operands.
"""
import pytest
import torch

from oracle import make_golden as G
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig
from tortoise_tts_amd.schedule import Schedule
from tests.gpu_util import quantize_sd

pytestmark = pytest.mark.gpu

@torch.no_grad()
def test_ar_step_graph_key_excludes_seed_row_offset_and_code_buffer():
    """One capture serves calls that differ in seed, row_offset (the candidate range of a rank / of a batch of the reference's
    autoregressive_batch_size loop) and receiving buffer; eager launches agree call by call."""
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(quantize_sd(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), torch.bfloat16), cfg)
    cond, text = G.ar_inputs(cfg)
    st = stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=8, max_text=40, max_new_tokens=32, max_latent_candidates=1)
    calls = [(5, 0), (5, 8), (6, 8), (5, 0), (7, 16)]
    got = []
    for seed, off in calls:
        st.prefill(cond, text)
        got.append(st.generate(8, 12, seed=seed, row_offset=off)[0].clone())
    assert st.stat(0) == 1, f"{st.stat(0)} captures for calls that differ in seed / row_offset / receiving buffer only"
    assert torch.equal(got[0], got[3]) and not torch.equal(got[0], got[1]) and not torch.equal(got[1], got[2])
    E.load_library().tt_graph_replay(0)
    try:
        for (seed, off), g in zip(calls, got):
            st.prefill(cond, text)
            assert torch.equal(st.generate(8, 12, seed=seed, row_offset=off)[0], g), f"kept graph differs from eager launches (seed {seed}, row_offset {off})"
    finally:
        E.load_library().tt_graph_replay(1)
    # rows [8, 16) of a 16-row call == an 8-row call at row_offset 8 (Philox streams are keyed by the global candidate index)
    st16 = stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=16, max_text=40, max_new_tokens=32, max_latent_candidates=1)
    st16.prefill(cond, text)
    both = st16.generate(16, 12, seed=5)[0]
    assert torch.equal(both[8:], got[1]) and torch.equal(both[:8], got[0])
    st16.close()
    st.close()


@torch.no_grad()
def test_diffusion_step_graph_is_kept_between_calls():
    """The sampler step is captured once per geometry: later calls with other noise / output tensors (other addresses) replay it through
    the handle's pointer table and equal eager launches bit for bit; another length re-captures."""
    cfg = DiffusionConfig(**G.DIFF_CFG)
    sd = quantize_sd(W.synthetic_state_dict(W.diffusion_manifest(cfg), seed=G.DIFF_SEED), torch.float16)
    S, latents, cond, x, step_noise = G.diff_inputs(cfg)
    st = stages.DiffusionStage(sd, cfg, dtype=E.TT_F16, max_seq=128, max_codes=64, max_steps=16)
    sched = Schedule(G.DIFF_STEPS, 4000, True, 2.0)
    g = torch.Generator().manual_seed(77)
    noises = [step_noise, torch.randn(step_noise.shape, generator=g), torch.randn(step_noise.shape, generator=g)]
    mels = []
    keep = []  # hold the earlier tensors so that the allocator hands out new addresses
    for nz in noises:
        st.condition(latents, cond, S)
        mel = st.sample(sched, x, nz)
        keep.append(mel)
        mels.append(mel.clone())
    assert st.stat(0) == 1, f"{st.stat(0)} captures for three calls of one geometry"
    assert not torch.equal(mels[0], mels[1])
    E.load_library().tt_graph_replay(0)
    try:
        for nz, m in zip(noises, mels):
            st.condition(latents, cond, S)
            assert torch.equal(st.sample(sched, x, nz), m), "the kept sampler-step graph differs from eager launches"
    finally:
        E.load_library().tt_graph_replay(1)
    S2 = S - 8
    st.condition(latents, cond, S2)
    mel_s = st.sample(sched, x[..., :S2].contiguous(), step_noise[..., :S2].contiguous())
    assert st.stat(0) == 2 and mel_s.shape[-1] == S2 and torch.isfinite(mel_s).all()
    st.condition(latents, cond, S)
    assert torch.equal(st.sample(sched, x, noises[0]), mels[0]) and st.stat(0) == 3
    assert st.guard() == 0
    st.close()


@torch.no_grad()
def test_overflow_guards_trip_on_fp16_overflow_only():
    """fp16 operands saturate at 65504.  Weights scaled so that an intermediate exceeds that: the stage's guard counts it with fp16
    operands and stays at zero with bf16 operands (and with the unscaled weights in fp16)."""
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), cfg)
    hot = dict(sd)
    for k in list(hot):
        if k.endswith("mlp.c_fc.weight") or k.endswith("mlp.c_fc.bias"):
            hot[k] = hot[k] * 1.0e5   # gelu(c_fc(h)) ~ 4e5: beyond the fp16 range, nothing for bf16
        if k.endswith("mlp.c_proj.weight"):
            hot[k] = hot[k] * 1.0e-5  # (keeps the residual stream tame in bf16)
    cond, text = G.ar_inputs(cfg)
    for weights, dt, want_trip in ((sd, E.TT_F16, False), (hot, E.TT_F16, True), (hot, E.TT_BF16, False)):
        st = stages.ArStage(weights, cfg, dtype=dt, max_batch=8, max_text=40, max_new_tokens=16, max_latent_candidates=1)
        st.prefill(cond, text)
        st.generate(8, 8, seed=1)
        n = st.guard()
        print(f"[guard] AR dtype={E.DTYPE_NAMES[dt]} scaled={weights is hot}: {n}")
        assert (n > 0) == want_trip, f"AR guard count {n} with dtype {E.DTYPE_NAMES[dt]} (scaled weights: {weights is hot})"
        assert st.guard() == 0, "guard(reset=True) did not clear the counter"
        st.close()
    dcfg = DiffusionConfig(**G.DIFF_CFG)
    dsd = W.synthetic_state_dict(W.diffusion_manifest(dcfg), seed=G.DIFF_SEED)
    dhot = dict(dsd)
    k_in = [k for k in dhot if k.endswith("in_layers.2.weight")][-1]  # a ResBlock's 1x1 convolution: its output feeds a GroupNorm
    dhot[k_in] = dhot[k_in] * 1.0e6
    S, latents, cond_d, x, step_noise = G.diff_inputs(dcfg)
    sched = Schedule(G.DIFF_STEPS, 4000, True, 2.0)
    for weights, dt, want_trip in ((dsd, E.TT_F16, False), (dhot, E.TT_BF16, False)):
        st = stages.DiffusionStage(weights, dcfg, dtype=dt, max_seq=128, max_codes=64, max_steps=16)
        st.condition(latents, cond_d, S)
        mel = st.sample(sched, x, step_noise)
        assert torch.isfinite(mel).all()
        n = st.guard()
        print(f"[guard] diffusion dtype={E.DTYPE_NAMES[dt]} scaled={weights is dhot}: {n}")
        assert (n > 0) == want_trip
        st.close()


@torch.no_grad()
def test_tts_demotes_an_overflowing_fp16_stage_to_bf16():
    """TextToSpeech with per-stage operand types: the diffusion stage (fp16 by default) is given weights whose x-path overflows fp16;
    the guard trips, the stage is rebuilt with bf16 operands and the utterance rendered again - finite audio, one demotion on record."""
    from tortoise_tts_amd.api import TextToSpeech, resolve_stage_dtypes
    assert resolve_stage_dtypes(None, False) == {"ar": E.TT_F16, "clvp": E.TT_F16, "diffusion": E.TT_F16, "vocoder": E.TT_F16}
    ar, clvp, diff = ARConfig(**G.AR_CFG), CLVPConfig(**G.CLVP_CFG), DiffusionConfig(**G.DIFF_CFG)
    dsd = dict(W.synthetic_state_dict(W.diffusion_manifest(diff), seed=G.DIFF_SEED))
    # inp_block scaled so that the integrating conv's operand inp_block(x) ~ 1e6 overflows fp16 while bf16 holds it
    dsd["inp_block.weight"] = dsd["inp_block.weight"] * 3.0e5
    dsd["integrating_conv.weight"] = dsd["integrating_conv.weight"] * 1.0e-5
    sds = {"autoregressive": W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(ar), seed=G.AR_SEED), ar),
           "clvp": W.synthetic_state_dict(W.clvp_manifest(clvp), seed=G.CLVP_SEED),
           "diffusion": dsd,
           "vocoder": W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(VocoderConfig()), seed=G.VOC_SEED))}
    tts = TextToSpeech(state_dicts=sds, configs={"ar": ar, "clvp": clvp, "diffusion": diff}, max_candidates=8, max_mel_tokens=32)
    assert tts.dtype_names() == {"ar": "fp16", "clvp": "fp16", "diffusion": "fp16", "vocoder": "fp16"}
    g = torch.Generator().manual_seed(2)
    lat = (torch.randn(1, ar.model_dim, generator=g) * 0.5, torch.randn(1, 2 * diff.model_channels, generator=g) * 0.5)
    with pytest.warns(UserWarning, match="diffusion stage overflowed fp16"):
        wav = tts.tts(list(range(10, 25)), conditioning_latents=lat, num_autoregressive_samples=8, diffusion_iterations=4, max_mel_tokens=24,
                      use_deterministic_seed=3, verbose=False)
    assert tts.demotions == ["diffusion"] and tts.dtype_names()["diffusion"] == "bf16"
    assert torch.isfinite(wav).all() and wav.abs().max() <= 1.0
    # the next utterance runs on the rebuilt stage without another demotion
    wav2 = tts.tts(list(range(10, 25)), conditioning_latents=lat, num_autoregressive_samples=8, diffusion_iterations=4, max_mel_tokens=24,
                   use_deterministic_seed=3, verbose=False)
    assert tts.demotions == ["diffusion"] and torch.equal(wav, wav2)
    for st in (tts.ar, tts.clvp, tts.diffusion, tts.vocoder):
        st.close()


@torch.no_grad()
def test_two_engines_on_two_threads_render_what_they_render_alone():
    """Round 3 removed a decode / render overlap because two engines working from two threads showed "an intermittent clip difference".
    Root cause (round 4): the write-through split-K slab stores of the decode projections are not ordered by the kernel boundary
    while another queue keeps the memory system busy - the row norm behind them summed stale slab values now and then.  With plain
    slab stores two handles on two threads / streams produce, run after run, exactly what each produces alone."""
    import threading
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), cfg)
    cond, text = G.ar_inputs(cfg)
    dcfg = DiffusionConfig(**G.DIFF_CFG)
    dsd = W.synthetic_state_dict(W.diffusion_manifest(dcfg), seed=G.DIFF_SEED)
    S, latents, dcond, x, step_noise = G.diff_inputs(dcfg)
    sched = Schedule(G.DIFF_STEPS, 4000, True, 2.0)
    B, steps, runs = 64, 40, 50
    ars = [stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=B, max_text=40, max_new_tokens=48, max_latent_candidates=1) for _ in range(2)]
    df = stages.DiffusionStage(dsd, dcfg, dtype=E.TT_F16, max_seq=128, max_codes=64, max_steps=16)
    ars[0].prefill(cond, text)
    want_codes = ars[0].generate(B, steps, seed=21)[0].clone()
    df.condition(latents, dcond, S)
    want_mel = df.sample(sched, x, step_noise).clone()
    errors = []

    def decode(st, seed_runs):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for r in range(seed_runs):
                    st.prefill(cond, text)
                    got = st.generate(B, steps, seed=21)[0]
                    if not torch.equal(got, want_codes):
                        errors.append(f"decode run {r}: {int((got != want_codes).any(dim=1).sum())} rows differ")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def render(seed_runs):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for r in range(seed_runs):
                    df.condition(latents, dcond, S)
                    got = df.sample(sched, x, step_noise)
                    torch.cuda.current_stream().synchronize()
                    if not torch.equal(got, want_mel):
                        errors.append(f"render run {r}: mel differs")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=decode, args=(ars[0], runs)), threading.Thread(target=decode, args=(ars[1], runs)),
               threading.Thread(target=render, args=(runs,))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:5]
    for st in ars:
        st.close()
    df.close()


@torch.no_grad()
def test_overlapped_integrator_prepass_is_scheduling_only():
    """Full-width denoiser, S = 870, 40 iterations = 3 chunks of the conditioning-integrator pre-pass: with the chunks on their own
    stream next to the sampler loop (the default) the mel is bit-identical to the whole pre-pass running first on the one stream,
    replayed or eager, run after run."""
    cfg = DiffusionConfig()
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), 1236)
    M, iters = 200, 40
    S = M * 4 * 24000 // 22050
    st = stages.DiffusionStage(sd, cfg, dtype=E.TT_F16, max_seq=S + 8, max_codes=M + 8, max_steps=64)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, M, 1024, generator=g)
    dcond = torch.randn(1, 2048, generator=g) * 0.5
    sched = Schedule(iters, cfg.trained_steps, True, 2)
    x = torch.randn(1, 100, S, generator=g)
    noise = torch.randn(iters, 1, 100, S, generator=g)
    st.set_option(E.TT_DIFF_OPT_OVERLAP_PREPASS, 0)
    st.condition(lat, dcond, S)
    want = st.sample(sched, x, noise).clone()
    assert torch.isfinite(want).all() and st.guard() == 0
    st.set_option(E.TT_DIFF_OPT_OVERLAP_PREPASS, 1)
    for rep in range(4):
        st.condition(lat, dcond, S)
        assert torch.equal(st.sample(sched, x, noise), want), f"the overlapped pre-pass changed the mel (repetition {rep})"
    E.load_library().tt_graph_replay(0)
    try:
        st.condition(lat, dcond, S)
        assert torch.equal(st.sample(sched, x, noise), want), "eager launches with the overlapped pre-pass changed the mel"
    finally:
        E.load_library().tt_graph_replay(1)
    assert st.stat(0) == 1
    st.close()


@pytest.mark.parametrize("dtype", [E.TT_F16, E.TT_BF16])
@torch.no_grad()
def test_fused_groupnorm_a_path_matches_the_standalone_apply(dtype):
    """ResBlock in_layers as one launch (TT_DIFF_OPT_FUSED_GN: GroupNorm32 + SiLU applied on the 1x1 conv's A path, csrc/gemm_gna.h)
    against the stand-alone apply launch, full-width denoiser at S = 870 (two samples: the tile at rows 832 .. 895 straddles them).
    Not bit-identical by construction - the fused path folds mean / rstd / gamma / beta into one multiply-add and takes the
    hardware reciprocal in the SiLU - but far inside the operand rounding: one eps evaluation and a 30-iteration mel, run twice."""
    cfg = DiffusionConfig()
    sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), 1237)
    M, iters = 200, 30
    S = M * 4 * 24000 // 22050
    st = stages.DiffusionStage(sd, cfg, dtype=dtype, max_seq=S + 8, max_codes=M + 8, max_steps=64)
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(1, M, 1024, generator=g)
    dcond = torch.randn(1, 2048, generator=g) * 0.5
    sched = Schedule(iters, cfg.trained_steps, True, 2)
    x = torch.randn(1, 100, S, generator=g)
    noise = torch.randn(iters, 1, 100, S, generator=g)
    st.set_option(E.TT_DIFF_OPT_FUSED_GN, 0)
    st.condition(lat, dcond, S)
    want = st.sample(sched, x, noise).clone()
    assert torch.isfinite(want).all() and st.guard() == 0
    st.set_option(E.TT_DIFF_OPT_FUSED_GN, 1)
    got = []
    for rep in range(2):
        st.condition(lat, dcond, S)
        got.append(st.sample(sched, x, noise).clone())
        assert torch.isfinite(got[-1]).all() and st.guard() == 0
    assert torch.equal(got[0], got[1]), "the fused path is not deterministic"
    rel = float((got[0] - want).norm() / want.norm())
    mx = float((got[0] - want).abs().max())
    print(f"[parity] fused GroupNorm A path vs stand-alone apply ({E.DTYPE_NAMES[dtype]}, S={S}, {iters} iterations): rel-L2 {rel:.3e} max-abs {mx:.3e}")
    assert rel < (4e-3 if dtype == E.TT_F16 else 3e-2), rel
    st.close()
