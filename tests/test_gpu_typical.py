"""-m gpu: typical sampling - tts(typical_sampling=True, typical_mass=...) (reference api.py:361-364; inference_speech hands
[TypicalLogitsWarper(mass)] to generate() as its logits_processor list, autoregressive.py:558; the warper itself is
tortoise/utils/typical_sampling.py:11-33) - on the device: the mask kernel against the committed outputs of the reference's own class
(tests/golden/typical.npz) and against the oracle with a repetition penalty in front, the sampler behind it against the oracle's
warped distribution, and the whole KV-cached decode loop (step graph included) against oracle.ar_sample_loop on shared Exp(1) draws.

Bar: the kept SET is integer work - token for token equal to the reference's, except at a float tie at the set's boundary (see
compare_sets: the reference's float32 entropy sum and its float32-rounded cumulative probability decide the last token; a row may
differ there by the boundary token, never elsewhere); such rows are counted and must stay rare.  The committed golden rows are equal
token for token."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as G
from oracle import tortoise_oracle as O
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig
from tests.gpu_util import quantize_sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return E.init()


def seen_mask(ids, V):
    m = np.zeros((ids.shape[0], (V + 31) // 32), dtype=np.uint32)
    for b in range(ids.shape[0]):
        for t in ids[b].tolist():
            m[b, t >> 5] |= np.uint32(1 << (t & 31))
    return torch.from_numpy(m.view(np.int32)).cuda()


def device_mask(lib, logits, seen, penalty, mass):
    B, V = logits.shape
    x = logits.cuda().contiguous()
    out = torch.full((B, V), 7.0, device="cuda")
    E.check(lib.tt_op_typical_mask(E.ptr(x), V, B, V, E.ptr(seen), penalty, mass, E.ptr(out), None))
    torch.cuda.synchronize()
    out = out.cpu()
    kept = out > -float("inf")
    assert torch.equal(out[kept], logits[kept]), "a kept token must carry its RAW logit (the sampler applies the penalty itself)"
    return kept


def distances64(scores):
    """|surprisal - entropy| per token from the float64 restatement."""
    logp = torch.log_softmax(scores.double(), -1)
    p = logp.exp()
    ent = -(logp * p).nansum(-1, keepdim=True)
    return ((-logp) - ent).abs(), p


def compare_sets(name, got, want, scores, mass):
    """Kept sets token for token.  A row may differ only AT the boundary: by at most two tokens whose distance lies within 2e-5 of the
    boundary distance tau, or - when the deciding cumulative sits within 2e-6 of the mass - by the boundary group itself.  Why 2e-5: the
    reference forms the entropy as a float32 sum of 8194 terms in ATen's vectorised order (observed error: a few 1e-6 at H ~ 9), the
    kernel accumulates it in double; an entropy that moves by e shifts the tokens above it against the tokens below it by 2 e, so a
    pair on opposite sides of the entropy whose distances differ by less than that may swap places at the boundary.  No device order can
    reproduce a CPU-vector-width-dependent sum; the kernel's own result is the exactly rounded one."""
    d, p = distances64(scores)
    differ = (got != want).any(1)
    n_tie = 0
    for r in differ.nonzero().flatten().tolist():
        finite = torch.isfinite(scores[r])
        tau = d[r][want[r] & finite].max()
        moved = (got[r] != want[r]).nonzero().flatten()
        ds, order = torch.sort(d[r])
        margin = float((p[r][order].cumsum(0) - mass).abs().min())
        near = bool(((d[r][moved] - tau).abs() < 2e-5).all())
        assert len(moved) <= 2 and (near or margin < 2e-6), \
            f"{name}: row {r} differs away from the boundary: tokens {moved.tolist()}, distances {d[r][moved].tolist()} vs tau {float(tau)}, margin {margin:.2e}"
        n_tie += 1
    print(f"[parity] typical mask {name}: rows {got.shape[0]}, kept min / max {int(want.sum(1).min())} / {int(want.sum(1).max())}, "
          f"rows differing {int(differ.sum())} (all at a float tie of the boundary: {n_tie})")
    return differ.nonzero().flatten().tolist()


@pytest.mark.parametrize("seed,scale,mass", G.TYPICAL_WARP_CASES)
def test_mask_equals_reference_warper_golden(lib, seed, scale, mass):
    """The device mask against the kept sets the reference's TypicalLogitsWarper produced (committed golden): 8194-wide rows with the
    stop token suppressed, a 3900-token band of -inf, an exact tie; no repetition penalty (the warper's own input)."""
    x = G.typical_warp_scores(seed, scale)
    want = torch.from_numpy(np.unpackbits(np.load(os.path.join(GOLD, "typical.npz"))[f"kept_s{seed}"], axis=1)[:, :G.TYPICAL_VOCAB].astype(bool))
    none_seen = torch.zeros(x.shape[0], (x.shape[1] + 31) // 32, dtype=torch.int32, device="cuda")
    got = device_mask(lib, x, none_seen, 1.0, mass)
    assert compare_sets(f"golden seed {seed} scale {scale} mass {mass}", got, want, x, mass) == []


def test_mask_behind_the_repetition_penalty_and_sampler_behind_the_mask(lib):
    """generate()'s order (4.31): RepetitionPenalty -> [caller's logits_processor = Typical] -> temperature -> top-k -> top-p.  The mask
    kernel applies the penalty for its decision; the sampler then draws from the oracle's warped distribution token for token."""
    g = torch.Generator().manual_seed(21)
    B, V = 64, 8194
    logits = torch.randn(B, V, generator=g) * torch.linspace(0.3, 6.0, B)[:, None]
    logits[:, 8193] = -float("inf")
    ids = torch.randint(0, V - 1, (B, 40), generator=g)
    ids[:, 0], ids[:, 1] = 1, 8192
    seen = seen_mask(ids, V)
    tie_rows = {}
    for mass in (0.9, 0.5, 0.2):
        pen = O.repetition_penalty_(logits.clone(), ids, 2.0)
        want = O.typical_(pen, mass) > -float("inf")
        got = device_mask(lib, logits, seen, 2.0, mass)
        tie_rows[mass] = compare_sets(f"penalty 2.0 mass {mass}", got, want, pen, mass)
    assert sum(len(v) for v in tie_rows.values()) <= 8  # 192 rows: boundary swaps must stay the exception
    # the sampler behind the mask: multinomial == argmax(p / q) on injected Exp(1) draws
    mass = 0.9
    q = torch.empty(1, B, V).exponential_(1, generator=g)
    scores = O.warp_logits(logits, ids, 2.0, 0.8, 50, 0.8, typical_mass=mass)
    want_tok = O.multinomial_from_exponential(torch.softmax(scores, -1), q[0])
    s = E.Sampling()
    s.temperature, s.top_p, s.repetition_penalty, s.top_k, s.seed, s.row_offset, s.typical_mass = 0.8, 0.8, 2.0, 50, 0, 0, mass
    qd = q.cuda().contiguous()
    s.exp_noise = E.ptr(qd)
    unfinished = torch.ones(B, dtype=torch.int32, device="cuda")
    codes = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
    ld = logits.cuda().contiguous()
    E.check(lib.tt_op_sample(E.ptr(ld), V, B, V, E.ptr(seen.clone()), C.byref(s), 0, E.ptr(unfinished), 8193, E.ptr(codes), 4, None))
    got_tok = codes[:, 0].cpu().long()
    clean = torch.tensor([r not in tie_rows[mass] for r in range(B)])  # (a row whose kept set moved at a float tie may draw another token)
    agree = float((got_tok == want_tok)[clean].float().mean())
    plain = O.multinomial_from_exponential(torch.softmax(O.warp_logits(logits, ids, 2.0, 0.8, 50, 0.8), -1), q[0])
    print(f"[parity] sampler behind the typical mask (mass {mass}): tokens equal to the oracle's {agree:.3f} on {int(clean.sum())} rows; "
          f"rows where typical sampling changes the token: {int((want_tok != plain).sum())}/{B}")
    assert agree == 1.0 and int((want_tok != plain).sum()) > 0
    assert torch.isfinite(scores[torch.arange(B), got_tok])[clean].all()


def test_mask_over_a_thousand_rows(lib):
    """How often the kept set differs from the reference's arithmetic at all: 1024 rows from near-uniform (scale 0.2: ~7 000 tokens kept)
    to peaked (scale 8: a handful), mass 0.9, repetition penalty 2.0 over 60 seen ids per row.  Every difference must be a boundary swap
    (compare_sets asserts it); the rate is printed and bounded."""
    g = torch.Generator().manual_seed(77)
    B, V = 1024, 8194
    logits = torch.randn(B, V, generator=g) * torch.logspace(-0.7, 0.9, B)[:, None]
    logits[:, 8193] = -float("inf")
    ids = torch.randint(0, V - 1, (B, 60), generator=g)
    seen = seen_mask(ids, V)
    pen = O.repetition_penalty_(logits.clone(), ids, 2.0)
    want = O.typical_(pen, 0.9) > -float("inf")
    got = device_mask(lib, logits, seen, 2.0, 0.9)
    rows = compare_sets("1024 rows, penalty 2.0, mass 0.9", got, want, pen, 0.9)
    moved = int((got != want).sum())
    print(f"[parity] typical mask over 1024 rows: {len(rows)} rows differ ({moved} of {int(want.sum())} kept tokens), each by its boundary token")
    assert len(rows) <= 20


def oracle_replay(st, cfg, cond, text, codes, noise, mass):
    """The oracle's warpers + draw applied to the ENGINE's own logits, step by step (teacher-forced with the engine's codes): the
    fraction of (row, step) pairs where that reproduces the engine's token.  Free-running codes cannot be compared with an fp32 CPU
    loop here: on these near-uniform synthetic distributions the typical set's BOUNDARY tokens are exactly the highest-scoring kept
    ones - what top-k then selects - so operand rounding of the logits moves the nucleus itself."""
    B, steps = codes.shape
    st.prefill(cond, text)
    lg = st.logits(1).cpu().expand(B, -1)
    st.begin(B)
    ids = torch.full((B, st.P + 1), 1, dtype=torch.long)
    ids[:, -1] = cfg.start_mel_token
    same = 0
    for s_ in range(steps):
        scores = O.warp_logits(lg, ids, 2.0, 0.8, 50, 0.8, typical_mass=mass)
        tok = O.multinomial_from_exponential(torch.softmax(scores, -1), noise[s_])
        same += int((tok == codes[:, s_]).sum())
        ids = torch.cat([ids, codes[:, s_:s_ + 1]], dim=1)
        if s_ + 1 < steps:
            st.decode_step(codes[:, s_])
            lg = st.logits(B).cpu()
    return same / (B * steps)


@pytest.mark.parametrize("mass", [0.9, 0.3])
@torch.no_grad()
def test_ar_generate_with_typical_sampling(mass):
    """The KV-cached decode loop with the typical mask in every step (first token from the shared prefill row, the rest inside the
    replayed step graph).  oracle.warp_logits(typical_mass) - pinned against real HF generate() runs through the reference's
    inference_speech(typical_sampling=True) (tests/golden/typical.npz) - applied to the engine's own teacher-forced logits with the
    same Exp(1) draws must reproduce every sampled token.  Then the properties the plain sampler has: reproducible, sharding-invariant
    Philox streams, one captured graph per setting, groups, and the fp32-mode loop against the fp32 oracle loop."""
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(quantize_sd(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), torch.bfloat16), cfg)
    cond, text = G.ar_inputs(cfg)
    B, steps = 4, 12
    gen = torch.Generator().manual_seed(3)
    noise = torch.empty(steps, B, cfg.number_mel_codes).exponential_(1, generator=gen)
    want = O.ar_sample_loop(sd, cfg, cond, text, B, steps, noise, typical_mass=mass)
    plain = O.ar_sample_loop(sd, cfg, cond, text, B, steps, noise)
    st32 = stages.ArStage(sd, cfg, dtype=E.TT_F32, max_batch=8, max_text=40, max_new_tokens=32, max_latent_candidates=1)
    st32.prefill(cond, text)
    got32, _ = st32.generate(B, steps, exp_noise=noise, typical_mass=mass)
    rows32 = float((got32.cpu() == want).all(1).float().mean())
    st32.close()
    st = stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=8, max_text=40, max_new_tokens=32, max_latent_candidates=1)
    st.prefill(cond, text)
    got, n = st.generate(B, steps, exp_noise=noise, typical_mass=mass)
    replay = oracle_replay(st, cfg, cond, text, got.cpu(), noise, mass)
    print(f"[parity] AR decode loop with typical sampling (mass {mass}): oracle warpers on the engine's own logits reproduce {replay:.3f} of the "
          f"sampled tokens; fp32-mode free-running rows equal to the fp32 oracle loop: {rows32:.2f}; tokens the option changes: "
          f"{int((want != plain).sum())}/{B * steps}")
    assert n == steps and replay == 1.0 and not torch.equal(want, plain)
    assert rows32 >= 0.5
    caps = st.lib.tt_ar_stat(st.h, 0)
    st.prefill(cond, text)
    again, _ = st.generate(B, steps, exp_noise=noise, typical_mass=mass)
    assert torch.equal(again, got) and st.lib.tt_ar_stat(st.h, 0) == caps  # the kept step graph is replayed
    st.prefill(cond, text)
    off, _ = st.generate(B, steps, exp_noise=noise)
    assert not torch.equal(off, got) and st.lib.tt_ar_stat(st.h, 0) == caps + 1  # another setting: another capture
    # Philox draws: reproducible, keyed by the global candidate index
    st.prefill(cond, text)
    full, _ = st.generate(4, steps, seed=9, typical_mass=mass)
    st.prefill(cond, text)
    hi, _ = st.generate(2, steps, seed=9, row_offset=2, typical_mass=mass)
    assert torch.equal(full[2:], hi)
    # several utterances in one decode batch (tts_many): each group's first token comes from ITS prefill row through the mask
    st.close()
    st2 = stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=8, max_text=40, max_new_tokens=32, max_latent_candidates=1, max_groups=2)
    text_b = torch.roll(text, 1, dims=1)
    alone = []
    for t_ in (text, text_b):
        st2.prefill(cond, t_)
        alone.append(st2.generate(4, steps, seed=9, typical_mass=mass)[0])
    st2.prefill_group(0, 2, cond, text)
    st2.prefill_group(1, 2, cond, text_b)
    both, _ = st2.generate(8, steps, seed=9, typical_mass=mass, group_seeds=[9, 9])
    assert torch.equal(both[:4], alone[0]) and torch.equal(both[4:], alone[1])
    with pytest.raises(E.EngineError, match="typical_mass"):
        st2.prefill(cond, text)
        st2.generate(4, steps, typical_mass=1.5)
    st2.close()


@torch.no_grad()
def test_streaming_chunks_with_typical_sampling_equal_one_shot():
    """tt_ar_generate_chunk (api_fast's loop) under typical sampling: resumed chunks == the one-shot generation."""
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(quantize_sd(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), torch.float16), cfg)
    cond, text = G.ar_inputs(cfg)
    st = stages.ArStage(sd, cfg, dtype=E.TT_F16, max_batch=1, max_text=40, max_new_tokens=40, max_latent_candidates=1)
    st.prefill(cond, text)
    one, n1 = st.generate(1, 30, seed=5, typical_mass=0.5)
    st.prefill(cond, text)
    last = None
    for codes, done in st.generate_stream(1, 30, 4, first_chunk=7, seed=5, typical_mass=0.5):
        assert torch.equal(codes, one[:, :codes.shape[1]])
        last = codes
    assert last.shape[1] == n1
    st.close()
