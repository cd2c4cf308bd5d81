"""CPU tests of the host-side logic that surrounds the HIP path: integer post-processing (bit-exact),
schedule tables, interpolation indices, weight re-layouts, the oracle's HF-sampling restatement."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tortoise_oracle as O
from tortoise_tts_amd import pack
from tortoise_tts_amd.schedule import Schedule, space_timesteps
from tortoise_tts_amd.stages import nearest_interp_index

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _api_helpers():
    # api.py imports the engine binding (ctypes only) -- importable without a GPU
    from tortoise_tts_amd import api
    return api


def test_fix_autoregressive_output_matches_reference_golden():
    api = _api_helpers()
    g = np.load(os.path.join(GOLD, "integer.npz"))
    got = api.fix_autoregressive_output(torch.from_numpy(g["codes_in"]), 8193)
    assert np.array_equal(got.numpy(), g["codes_out"])
    # edge cases: no stop token (unchanged), stop in the last three slots, all stop
    rows = torch.tensor([[5, 6, 7, 8, 9, 10], [5, 6, 7, 8, 8193, 10], [8193] * 6, [1, 2, 3, 4, 5, 8193]])
    want = np.stack([O.fix_autoregressive_output(r.numpy(), 8193) for r in rows])
    assert np.array_equal(api.fix_autoregressive_output(rows, 8193).numpy(), want)


def test_fix_autoregressive_output_randomised_ragged_rows():
    """Ragged batches: stop tokens at random positions (including 0, the last three slots, repeated, absent);
    the batched device-side form must equal the row-wise restatement of api.py:87-114 bit for bit."""
    api = _api_helpers()
    rng = np.random.default_rng(7)
    for trial in range(300):
        B, n = int(rng.integers(1, 9)), int(rng.integers(3, 40))
        rows = rng.integers(0, 8192, (B, n))
        for b in range(B):
            for _ in range(int(rng.integers(0, 4))):
                rows[b, int(rng.integers(0, n))] = 8193
        want = np.stack([O.fix_autoregressive_output(r, 8193) for r in rows])
        got = api.fix_autoregressive_output(torch.from_numpy(rows), 8193).numpy()
        assert np.array_equal(got, want), rows


def test_calm_trim_matches_oracle():
    api = _api_helpers()
    rng = np.random.default_rng(1)
    for trial in range(200):
        n = int(rng.integers(1, 60))
        row = rng.integers(80, 86, n)
        if trial % 2:
            s = int(rng.integers(0, n))
            row[s:s + int(rng.integers(1, 14))] = 83
        assert api.calm_trim_length(torch.from_numpy(row)) == O.calm_trim_length(row), row


@pytest.mark.parametrize("steps", [5, 30, 80, 200, 400])
def test_schedule_matches_oracle(steps):
    a, b = Schedule(steps), O.Schedule(steps)
    assert list(a.timestep_map) == list(b.timestep_map)
    for name in ("betas", "sqrt_recip_ac", "sqrt_recipm1_ac", "post_logvar_clipped", "log_betas", "coef1", "coef2"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert sorted(space_timesteps(4000, [5])) == [0, 1000, 2000, 2999, 3999]  # SURVEY.md §8a-9 [probed]


@pytest.mark.parametrize("m,s", [(12, 52), (200, 870), (500, 2176), (37, 161)])
def test_nearest_interp_index(m, s):
    src = torch.arange(m, dtype=torch.float32)[None, None]
    want = F.interpolate(src, size=s, mode="nearest")[0, 0].long().numpy()
    assert np.array_equal(nearest_interp_index(m, s), want)


def test_relpos_table_matches_oracle_bias():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(32, 4, generator=g)
    tab = pack.relpos_table(w, 8.0)  # [heads][129]
    n = 200
    bias = O.rel_pos_bias(w, n, 8.0)  # [heads][n][n], index [q][k]
    pos = torch.arange(n)
    d = (pos[None, :] - pos[:, None]).clamp(-64, 64) + 64
    assert torch.equal(tab[:, d], bias)


def test_qkv_permutation():
    C, H = 128, 2
    perm = pack._qkv_head_major_perm(C, H)
    assert sorted(perm.tolist()) == list(range(3 * C))
    old = torch.arange(3 * C)
    new = old[perm]
    # new layout [part][head][64]  <- old layout [head][part][64]
    assert new.reshape(3, H, 64)[1, 1, 5] == old.reshape(H, 3, 64)[1, 1, 5]
    assert new.reshape(3, H, 64)[2, 0, 63] == old.reshape(H, 3, 64)[0, 2, 63]


def test_sampling_restatement_against_installed_transformers():
    """The reference pins transformers==4.31 (not installable offline); the installed release still ships the
    same four logits processors, which is the strongest available pin for oracle.warp_logits."""
    tf = pytest.importorskip("transformers")
    try:
        from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                            TopKLogitsWarper, TopPLogitsWarper)
    except Exception:
        pytest.skip("logits processors not importable")
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 8194, generator=g) * 4
    ids = torch.randint(0, 8194, (5, 30), generator=g)
    want = RepetitionPenaltyLogitsProcessor(2.0)(ids, logits.clone())
    want = TemperatureLogitsWarper(0.8)(ids, want)
    want = TopKLogitsWarper(50)(ids, want)
    want = TopPLogitsWarper(0.8)(ids, want)
    got = O.warp_logits(logits, ids, 2.0, 0.8, 50, 0.8)
    assert torch.equal(torch.isfinite(got), torch.isfinite(want))
    m = torch.isfinite(want)
    assert torch.allclose(got[m], want[m], atol=1e-6)


def test_multinomial_equals_argmax_exponential():
    """torch.multinomial(p, 1) on CPU == argmax(p / q), q ~ Exp(1) from the same generator state."""
    for seed in range(10):
        p = torch.softmax(torch.randn(4, 8194, generator=torch.Generator().manual_seed(100 + seed)) * 3, -1)
        g1 = torch.Generator().manual_seed(seed)
        g2 = torch.Generator().manual_seed(seed)
        want = torch.multinomial(p, 1, generator=g1)[:, 0]
        q = torch.empty_like(p).exponential_(1, generator=g2)
        assert torch.equal(O.multinomial_from_exponential(p, q), want)


def test_split_and_recombine_text_reference_expectations():
    """The reference's own known-answer tests for long-form chunking (tortoise/utils/text.py:82-130), stored with
    their inputs in tests/golden/text_split.json by oracle/make_golden.py."""
    import json
    from tortoise_tts_amd.text import split_and_recombine_text
    with open(os.path.join(os.path.dirname(__file__), "golden", "text_split.json")) as f:
        cases = json.load(f)
    assert len(cases) == 3
    for c in cases:
        assert split_and_recombine_text(c["text"], c["desired_length"], c["max_length"]) == c["chunks"]
    assert len(cases[2]["chunks"]) == 15  # riding_hood.txt -> 15 utterances
    # edge cases: empty / whitespace / punctuation-only input, one over-long word, no terminal punctuation
    assert split_and_recombine_text("") == []
    assert split_and_recombine_text("  \n\n ... !!! ") == []
    assert split_and_recombine_text("word") == ["word"]
    # a word longer than max_length has no break to fall back to: it is cut back to desired_length pieces
    assert split_and_recombine_text("x" * 50, desired_length=10, max_length=20) == ["x" * 10] * 5
    for chunk in split_and_recombine_text("one two three four five six seven eight nine ten " * 20, 40, 60):
        assert 0 < len(chunk) <= 60


def test_split_and_recombine_text_against_reference_source():
    """Randomised cross-check against the reference function itself (build container only)."""
    import importlib.util
    import random
    from oracle import ref_shims
    from tortoise_tts_amd.text import split_and_recombine_text
    path = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "utils", "text.py")
    if not os.path.exists(path):
        pytest.skip("reference tree not available")
    spec = importlib.util.spec_from_file_location("_ref_text", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = random.Random(1)
    words = ["alpha", "be", "c", "delta-epsilon", "z" * 44, "It's", "ok"]
    punct = [". ", "! ", "? ", "... ", "?! ", ", ", "; ", "\n", "\n\n", ' "', '" ', "\u201c", "\u201d ", " ", ".", '"']
    for _ in range(500):
        s = "".join(rng.choice(words) + rng.choice(punct) for _ in range(rng.randint(0, 60)))
        d = rng.choice([10, 20, 50, 200])
        m = d + rng.choice([1, 10, 30, 100])
        assert split_and_recombine_text(s, d, m) == ref.split_and_recombine_text(s, d, m), repr(s)


def test_tokenizer_contract_on_reference_vocabulary(monkeypatch):
    """VoiceBpeTokenizer over the vocabulary file the reference ships (tortoise/data/tokenizer.json is data the user
    already has; it is located, not copied - so this runs only where the reference tree is).  The do_tts.py default
    sentence gives 54 ids, spaces map to the [SPACE] token and decode() inverts encode()."""
    from oracle import ref_shims
    from tortoise_tts_amd.text import VoiceBpeTokenizer
    vocab = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "data", "tokenizer.json")
    if not os.path.exists(vocab):
        pytest.skip("reference vocabulary not available")
    tok = VoiceBpeTokenizer(vocab, use_basic_cleaners=True)
    text = "The expressiveness of autoregressive transformers is literally nuts! I absolutely adore them."
    ids = tok.encode(text)
    assert len(ids) == 54 and max(ids) < 255 and min(ids) >= 0
    space = tok.tokenizer.token_to_id("[SPACE]")
    assert ids.count(space) == text.count(" ")
    assert tok.decode(ids) == text.lower()
    assert tok.encode("  Mixed   CASE\tand\nwhitespace ") == tok.encode(" mixed case and whitespace ")
    # no explicit file, no environment override, no models_dir copy and no installed `tortoise` package to fall back on
    # (other tests of the session may have put the reference tree on sys.path)
    import importlib.util
    monkeypatch.delenv("TORTOISE_TOKENIZER", raising=False)
    monkeypatch.setattr(importlib.util, "find_spec", lambda name, *a, **k: None)
    with pytest.raises(FileNotFoundError):
        VoiceBpeTokenizer(os.path.join(os.path.dirname(vocab), "missing.json"), use_basic_cleaners=True, models_dir="/nonexistent")


def test_committed_bench_line_follows_the_contract():
    """The newest bench line committed under profiles/ carries every field the measurement contract names, the roofline
    fraction is achieved / peak, and roofline.traffic is what bench.pmc_traffic derives from the committed PMC pass of the
    same round (same kernel class name, same source digest => not stale)."""
    import glob
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sorted(glob.glob(os.path.join(root, "profiles", "r*_final_bench_1gpu.json")))[-1]
    with open(path) as f:
        d = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - d["audio_seconds_per_step"] / (d["ms_per_step"] / 1e3)) < 1e-6
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.pmc_traffic("no_such_kernel") == (None, None, None)
    if r.get("traffic_source", "").startswith(bench.PMC_SUMMARY):  # the line was produced against this round's PMC pass
        traffic, src, stale = bench.pmc_traffic(r["kernel"])
        assert src and abs(traffic - r["traffic"]) < 1.0
        assert "ar_step_s_by_threads" in c  # thread sweep of the CPU baseline is part of the line


def test_utterance_batch_capacity_arithmetic():
    """TextToSpeech(utterance_batch=) refuses what cannot fit the device: the arithmetic behind the message, at the reference sizes."""
    from tortoise_tts_amd.api import utterance_batch_bytes
    from tortoise_tts_amd.config import ARConfig, DiffusionConfig
    ar, df = ARConfig(), DiffusionConfig()
    per_token = ar.layers * ar.model_dim * 2 * 2  # K and V, 2 bytes each: 122 880 B per cached token
    assert per_token == 122880
    # the long-form benchmark: 16 utterances x 256 candidates x 200 tokens
    need = utterance_batch_bytes(16, 256, 200, ar, df)
    kv = 16 * 256 * 202 * per_token
    S = 200 * 4 * 24000 // 22050 + 8
    assert need == kv + 512 * 2 * 16 * S * df.model_channels * 2
    assert 100 * 2 ** 30 < need < 0.8 * 288 * 2 ** 30           # fits one MI355X
    assert utterance_batch_bytes(16, 256, 500, ar, df) > 0.8 * 288 * 2 ** 30   # the full 500-token capacity at 16 x 256 does not
    assert utterance_batch_bytes(1, 256, 500, ar, df) < 0.1 * 288 * 2 ** 30
