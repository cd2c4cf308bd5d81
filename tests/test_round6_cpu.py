"""CPU checks of round 6: the packed one-collective gather, `python bench.py --gpus N` starting its own ranks, and the oracle against the
long-context goldens of the reference's GPT2InferenceModel (tests/golden/full_ar_long.npz)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tortoise_tts_amd import dist as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("n_loc,M,ws", [(2, 8, 2), (3, 5, 3), (32, 200, 8), (1, 1, 4), (12, 501, 8)])
def test_packed_gather_payload_round_trips(n_loc, M, ws):
    """scores (f32, bit-cast) + int16 codes of every rank in ONE int32 buffer: packing then unpacking the concatenation of `ws` rank
    payloads returns every score bit for bit and every code, for odd n_loc * M as well (the padding half-word)."""
    g = torch.Generator().manual_seed(n_loc * 100 + M)
    scores = torch.randn(ws, n_loc, generator=g)
    scores[0, 0] = float("-inf")  # bit patterns, not values, travel
    codes = torch.randint(0, 8194, (ws, n_loc, M), generator=g, dtype=torch.int32)
    words = torch.cat([tdist.pack_candidates(scores[r], codes[r]) for r in range(ws)])
    assert words.dtype == torch.int32 and words.numel() == ws * (n_loc + (n_loc * M + 1) // 2)
    s_all, c_all = tdist.unpack_candidates(words, ws, n_loc, M)
    assert torch.equal(s_all.view(torch.int32), scores.reshape(-1).view(torch.int32))
    assert torch.equal(c_all, codes.reshape(ws * n_loc, M))
    with pytest.raises(ValueError):
        tdist.pack_candidates(scores[0], codes[0] + 40000)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment - the form the driver uses - spawns its two ranks itself
    (torch.distributed.run on 127.0.0.1), the ranks rendezvous, run the path's ONE all_gather (dist.gather_candidates) and rank 0 prints
    one JSON line.  --rank-check keeps the engines out (no GPU here); the GPU box runs the complete flow (scripts/gpu.sh ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["rank_check"] is True and d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["all_gather_calls"] == 1, d


def test_self_launch_command_is_the_drivers_form():
    import bench
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "3"], 8, 29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[4:6] == ["--nproc-per-node", "8"]
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")


@torch.no_grad()
def test_oracle_follows_the_reference_decode_beyond_the_first_key_block():
    """oracle.ar_step teacher-forced through 65 cached steps on the benchmark weights against the reference GPT2InferenceModel's logits
    after 1 / 63 / 64 / 65 fed tokens (tests/golden/full_ar_long.npz; 8 rows): pins the oracle's KV-cached position rule and attention at
    contexts well past the prefill (the 3-step golden full_ar.npz stops at context 62)."""
    import bench
    from oracle import make_golden_full as GF
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd.config import ARConfig
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = np.load(os.path.join(GOLD, "full_ar_long.npz"))
    cfg = ARConfig()
    sd = bench.synthetic_weights()["autoregressive"]
    text, auto, _ = GF.prompt()
    toks = GF.arl_tokens()
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False
    lg, kv = O.ar_prefill(sd, cfg, O.ar_prefix(sd, cfg, auto, text), GF.ARL_B)
    for s in range(65):
        lg, kv = O.ar_step(sd, cfg, toks[s], s + 1, kv)
        if s + 1 in (1, 63, 64, 65):
            want = torch.from_numpy(g["logits_%d" % (s + 1)])
            rel = float((lg[:, keep] - want[:, keep]).norm() / want[:, keep].norm())
            assert rel < 2e-5, (s + 1, rel)
