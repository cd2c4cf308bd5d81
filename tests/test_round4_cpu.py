"""Host logic added in round 4 (no GPU): per-stage operand types, the overflow-guard demotion path of TextToSpeech, bench.py's dtype
label, and the collective-initialisation fallback under the launcher bench.py prescribes (torch.distributed.run)."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from tests import fake_stages
from tortoise_tts_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_dtype_resolution():
    from tortoise_tts_amd.api import resolve_stage_dtypes
    bf, fp = E.TT_BF16, E.TT_F16
    # the reference: AR + CLVP under fp16 autocast only with half=True (api.py:413-414); diffusion + vocoder in fp32 always (api.py:225, 540-560)
    assert resolve_stage_dtypes(None, False) == {"ar": fp, "clvp": fp, "diffusion": fp, "vocoder": fp}  # round 6: fp16 + overflow guard everywhere
    assert resolve_stage_dtypes(None, True) == {"ar": fp, "clvp": fp, "diffusion": fp, "vocoder": fp}
    assert resolve_stage_dtypes("bf16", False) == {k: bf for k in ("ar", "clvp", "diffusion", "vocoder")}
    assert resolve_stage_dtypes({"diffusion": "bf16"}, False) == {"ar": fp, "clvp": fp, "diffusion": bf, "vocoder": fp}
    assert resolve_stage_dtypes({"ar": "fp16", "clvp": "f16"}, True)["ar"] == fp
    with pytest.raises(ValueError, match="half=True"):
        resolve_stage_dtypes("bf16", True)
    with pytest.raises(ValueError, match="unknown stage"):
        resolve_stage_dtypes({"vocoder2": "bf16"}, False)
    with pytest.raises(ValueError, match="unknown operand type"):
        resolve_stage_dtypes("fp8", False)


def test_bench_dtype_label():
    import bench
    assert bench.dtype_label({"ar": "bf16", "clvp": "bf16", "diffusion": "bf16", "vocoder": "bf16"}) == "bf16"
    assert bench.dtype_label({"ar": "bf16", "clvp": "bf16", "diffusion": "fp16", "vocoder": "fp16"}) == "bf16(ar,clvp)+fp16(diffusion,vocoder)"


@torch.no_grad()
def test_overflowing_fp16_stage_is_rebuilt_with_bf16_and_the_utterance_rendered_again(monkeypatch):
    fake_stages.install(monkeypatch)
    from tests.test_api_flow_cpu import small_setup, voice_latents
    from tortoise_tts_amd.api import TextToSpeech
    sds, cfgs = small_setup()
    tts = TextToSpeech(state_dicts=sds, configs=cfgs, max_candidates=8, max_mel_tokens=16, kv_cache=True)
    assert tts.dtype_names() == {"ar": "fp16", "clvp": "fp16", "diffusion": "fp16", "vocoder": "fp16"}
    lat = voice_latents(cfgs)
    kw = dict(conditioning_latents=lat, num_autoregressive_samples=4, diffusion_iterations=2, max_mel_tokens=10, use_deterministic_seed=4, verbose=False)
    clean = tts.tts(list(range(30, 40)), **kw)
    assert tts.demotions == []
    first = tts.diffusion
    fake_stages.FakeDiffusionStage.trip = 3  # the fp16 diffusion stage reports non-finite statistics once
    try:
        with pytest.warns(UserWarning, match="diffusion stage overflowed fp16"):
            again = tts.tts(list(range(30, 40)), **kw)
    finally:
        fake_stages.FakeDiffusionStage.trip = 0
    assert tts.demotions == ["diffusion"] and tts.dtype_names()["diffusion"] == "bf16"
    assert tts.diffusion is not first and getattr(first, "closed", False), "the overflowing stage was not rebuilt"
    assert torch.equal(clean, again)  # (the stand-in computes in fp32 either way: same seed, same audio)
    # a stage that reports non-finite values with bf16 operands is an error, not a precision choice
    monkeypatch.setattr(fake_stages.FakeDiffusionStage, "guard", lambda self, reset=True: 1)
    with pytest.raises(E.OperandOverflow):
        tts.tts(list(range(30, 40)), **kw)


def test_collective_init_failure_leaves_rank0_serving_under_torch_distributed_run(tmp_path):
    """bench.py's launcher: when the collectives cannot initialise, ranks != 0 must end with status 0 (a FAILED worker makes the
    elastic agent terminate the group, rank 0 included) and rank 0 carries on as a one-GPU engine."""
    script = tmp_path / "fallback_probe.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch.distributed as dist
        def boom(*a, **k):
            raise RuntimeError("simulated RCCL initialisation failure")
        dist.init_process_group = boom
        from tortoise_tts_amd import dist as tdist
        try:
            tdist.init_from_env()
            caught = False
        except Exception as ex:   # the library raises an ordinary exception (round 5): an embedding host can catch it ...
            caught = isinstance(ex, tdist.CollectiveInitFailed) and isinstance(ex, RuntimeError)
        assert caught == (int(os.environ["RANK"]) != 0)
        rank, world, local = tdist.init_from_env_or_exit()   # ... and the entry-point form makes ranks != 0 leave here with status 0
        assert (rank, world) == (0, 1) and tdist.FALLBACK_SINGLE
        assert tdist.any_over_ranks([True, False]) == [True, False] and tdist.max_over_ranks(2.5) == 2.5
        open(os.path.join({str(tmp_path)!r}, "rank0_served"), "w").write("ok")
    """))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "rank0_served").exists(), r.stderr[-2000:]
    assert r.stderr.count("falling back to single-GPU operation") == 3


@torch.no_grad()
def test_tts_many_decodes_again_when_the_fp16_autoregressive_stage_overflowed(monkeypatch):
    """Round-4 advisor finding: with utterance_batch == 1 tts_many decodes every utterance's candidates up front and hands them to tts()
    as '_ar_samples'; an overflowed fp16 decode (rows cut short with the stop token) was then rendered, and re-rendered after the
    demotion, from the SAME stale codes.  The guard is read right behind the decode now: the stage is rebuilt with bf16 operands and
    the texts are decoded again - same audio as an engine that ran bf16 from the start."""
    fake_stages.install(monkeypatch)
    from tests.test_api_flow_cpu import small_setup, voice_latents
    from tortoise_tts_amd.api import TextToSpeech
    sds, cfgs = small_setup()
    lat = voice_latents(cfgs)
    texts = [list(range(30, 40)), list(range(41, 49))]
    kw = dict(conditioning_latents=lat, num_autoregressive_samples=4, diffusion_iterations=2, max_mel_tokens=10, use_deterministic_seed=4, verbose=False)
    ref = TextToSpeech(state_dicts=sds, configs=cfgs, max_candidates=8, max_mel_tokens=16, kv_cache=True).tts_many(texts, **kw)
    tts = TextToSpeech(state_dicts=sds, configs=cfgs, max_candidates=8, max_mel_tokens=16, kv_cache=True, dtype={"ar": "fp16"})
    assert tts.dtype_names()["ar"] == "fp16"
    first = tts.ar
    fake_stages.FakeArStage.trip = 2
    try:
        with pytest.warns(UserWarning, match="ar stage overflowed fp16"):
            out = tts.tts_many(texts, **kw)
    finally:
        fake_stages.FakeArStage.trip = 0
    assert tts.demotions == ["ar"] and tts.dtype_names()["ar"] == "bf16"
    assert tts.ar is not first and getattr(first, "closed", False)
    assert len(out) == len(ref) and all(torch.equal(a, b) for a, b in zip(out, ref)), "rendered from the overflowed stage's codes"
