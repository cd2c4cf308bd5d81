"""-m gpu: CVVP scoring on the device (tt_cvvp_score, csrc/cvvp.hip) - the optional second ranking model of tts(cvvp_amount > 0)
(reference: tortoise/models/cvvp.py:63-131, driven per conditioning clip at api.py:464-472) - against the oracle on operand-rounded
weights and against the committed outputs of the reference's own CVVP class (tests/golden/cvvp.npz: a reduced instance and the
512-wide / 8-head / depth-8 instance api.py:254 builds).  Scores are cosine similarities x exp(temperature) (|score| < e): tolerances
are ABSOLUTE - fp32 verification mode 2e-5 (measured 3e-7), fp16 operands 2e-3 (measured 5e-4), bf16 operands 1.5e-2 (measured 3.8e-3) - plus the ranking (what api.py:477's top-k consumes)."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as G
from oracle import tortoise_oracle as O
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import CVVPConfig
from tests.gpu_util import quantize_sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = [("bf16", E.TT_BF16, torch.bfloat16, 1.5e-2), ("f16", E.TT_F16, torch.float16, 2e-3), ("f32", E.TT_F32, torch.float32, 2e-5)]


def spearman(a, b):
    ra = torch.argsort(torch.argsort(a.double())).double()
    rb = torch.argsort(torch.argsort(b.double())).double()
    ra, rb = ra - ra.mean(), rb - rb.mean()
    return float((ra * rb).sum() / (ra.norm() * rb.norm()))


@pytest.mark.parametrize("name,dt,tdt,tol", MODES)
@pytest.mark.parametrize("tag", ["small", "full"])
@torch.no_grad()
def test_cvvp_scores(tag, name, dt, tdt, tol):
    cfg = CVVPConfig(**G.CVVP_CFG) if tag == "small" else CVVPConfig()
    sd0 = W.synthetic_state_dict(W.cvvp_manifest(cfg), seed=G.CVVP_SEED)
    sd = quantize_sd(sd0, tdt)
    mels, codes = G.cvvp_inputs(tag == "full")
    st = stages.CvvpStage(sd, cfg, dtype=dt, max_rows=4096, max_cond_frames=160)
    got = st.score(mels, codes).cpu()
    want = O.cvvp_score(sd, cfg, mels, codes)
    ref = torch.from_numpy(np.load(os.path.join(GOLD, "cvvp.npz"))[f"scores_{tag}"])
    e_o, e_r = float((got - want).abs().max()), float((got - ref).abs().max())
    print(f"[parity] CVVP scores {tag} {name}: max abs vs oracle (same rounded weights) {e_o:.2e}, vs the reference class's golden {e_r:.2e} "
          f"(tol {tol:.0e} / {2 * tol:.0e}); scores {[round(v, 4) for v in got.tolist()]}")
    assert e_o < tol and e_r < 2 * tol
    assert st.guard() == 0
    # a second call on the same handle, other shapes (buffers are reused between the conditioning and the speech tower)
    again = st.score(mels, codes).cpu()
    assert torch.equal(again, got)
    one_clip = st.score(mels[:, :1], codes[:3]).cpu()
    assert float((one_clip - O.cvvp_score(sd, cfg, mels[:, :1], codes[:3])).abs().max()) < tol
    st.close()


@torch.no_grad()
def test_cvvp_ranks_a_full_candidate_batch_like_the_oracle():
    """The api.py:254 instance over 64 candidates x 200 codes and two 517-frame clips (what api.py:73-84's format_conditioning produces),
    chunked over two calls by a small max_rows: every score within the fp16 tolerance of the fp32 oracle, same top candidates."""
    cfg = CVVPConfig()
    sd = quantize_sd(W.synthetic_state_dict(W.cvvp_manifest(cfg), seed=G.CVVP_SEED), torch.float16)
    g = torch.Generator().manual_seed(23)
    mels = torch.randn(1, 2, 80, 517, generator=g) * 2 - 5
    codes = torch.randint(0, 8192, (64, 200), generator=g)
    st = stages.CvvpStage(sd, cfg, dtype=E.TT_F16, max_rows=32 * 200, max_cond_frames=520)
    got = st.score(mels, codes).cpu()
    want = O.cvvp_score(sd, cfg, mels, codes)
    err, rho = float((got - want).abs().max()), spearman(got, want)
    top_g, top_w = set(torch.topk(got, 8).indices.tolist()), set(torch.topk(want, 8).indices.tolist())
    print(f"[parity] CVVP 64 candidates x 200 codes, 2 clips x 517 frames (fp16): max abs {err:.2e}, Spearman {rho:.5f}, top-8 overlap {len(top_g & top_w)}/8, "
          f"score spread {float(want.std()):.3f}")
    assert err < 2e-3 and rho > 0.999 and len(top_g & top_w) >= 7
    st.close()


@torch.no_grad()
def test_tts_with_cvvp_amount_on_the_device():
    """The whole tts() call at the reference's hyper-parameters with a voice given as clips (mel pairs) and cvvp_amount in {0, 0.5, 1}: the CVVP
    stage is built on first use (api.py:234, 450-453), the ranking tts() used equals the blend api.py:462-472 prescribes - recomputed here
    from the stages' own scores of the same candidates - and cvvp_amount = 1 never runs CLVP."""
    import bench
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd.api import TextToSpeech, fix_autoregressive_output
    import torch.nn.functional as F
    sds = bench.synthetic_weights()
    sds["cvvp"] = W.synthetic_state_dict(W.cvvp_manifest(CVVPConfig()), seed=G.CVVP_SEED)
    text, _ = bench.synthetic_prompt()
    tts = TextToSpeech(state_dicts=sds, max_candidates=16, max_mel_tokens=48)
    g = torch.Generator().manual_seed(12)
    pairs = [(torch.randn(1, 80, 517, generator=g) * 2 - 5, torch.randn(1, 100, 564, generator=g) * 2 - 5) for _ in range(2)]
    kw = dict(voice_samples=pairs, num_autoregressive_samples=16, diffusion_iterations=4, max_mel_tokens=48, use_deterministic_seed=5, k=3, verbose=False)
    wav0 = tts.tts(text, **kw)
    assert tts.cvvp is None and len(wav0) == 3
    best0 = tts.last_best_codes.clone()
    calls = {"clvp": 0}
    orig = tts.clvp.score
    tts.clvp.score = lambda *a, **k_: (calls.__setitem__("clvp", calls["clvp"] + 1), orig(*a, **k_))[1]
    wav1 = tts.tts(text, cvvp_amount=1.0, **kw)
    assert tts.cvvp is not None and calls["clvp"] == 0 and all(torch.isfinite(w).all() for w in wav1)
    best1 = tts.last_best_codes.clone()
    tts.tts(text, cvvp_amount=0.5, **kw)
    assert calls["clvp"] == 1
    best_half = tts.last_best_codes.clone()
    # the same candidates again (same seed), scored by the two stages directly
    auto, _, auto_conds, _ = tts.get_conditioning_latents(pairs, return_mels=True)
    tt = F.pad(text.int()[None].to(tts.device), (0, 1))
    tts.ar.prefill(auto.to(tts.device).float(), tt)
    codes, _ = tts.ar.generate(16, 48, seed=5, row_offset=0)
    fixed = fix_autoregressive_output(F.pad(codes, (0, 48 - codes.shape[1]), value=tts.stop_mel_token), tts.stop_mel_token)
    clvp, cvvp = orig(tt, fixed), tts.cvvp.score(auto_conds, fixed)
    assert float(cvvp.std()) > 1e-3
    for amount, best in ((0.0, best0), (1.0, best1), (0.5, best_half)):
        scores = O.blend_candidate_scores(clvp, cvvp, amount)
        assert torch.equal(best.cpu(), fixed[torch.topk(scores, 3).indices].cpu()), f"ranking at cvvp_amount={amount}"
    print(f"[parity] tts(cvvp_amount): winners at 0 / 0.5 / 1 = {torch.topk(clvp, 3).indices.tolist()} / "
          f"{torch.topk(O.blend_candidate_scores(clvp, cvvp, 0.5), 3).indices.tolist()} / {torch.topk(cvvp, 3).indices.tolist()}; "
          f"CVVP score spread {float(cvvp.std()):.3f}, CLVP {float(clvp.std()):.3f}; stage seconds {({k: round(v, 4) for k, v in tts.timings.items()})}")
