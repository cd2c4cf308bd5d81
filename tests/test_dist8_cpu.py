"""World-size-8 `gloo` run of TextToSpeech.tts()'s multi-rank control flow on the CPU stand-ins (tests/fake_stages.py): what the
driver's 8-GPU tier executes and no 1-GPU box can - 2 candidates per rank, ONE all_gather of scores + codes, the identical top-k on
every rank, the pair group {0, 1} inside the 8-rank world rendering a single winner's split diffusion tail while ranks 2 - 7 skip
it, k = 3 winners rendered round-robin by ranks 0 / 1 / 2 and sent to rank 0, the agreed overflow-guard flags.  The audio must equal
the single-process rendering of the same seeded utterance (Philox / generator streams are keyed by the global candidate index)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tortoise_tts_amd import dist as tdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


KW = dict(num_autoregressive_samples=16, diffusion_iterations=3, max_mel_tokens=12, use_deterministic_seed=9, verbose=False)
TEXT = list(range(20, 34))


def _worker(rank, world, port, out_dir):
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if world > 1:
        tdist.init_from_env()
        tdist._PAIR = None
    from tests import fake_stages
    from tests.test_api_flow_cpu import small_setup, voice_latents
    mp_ = pytest.MonkeyPatch()
    fake_stages.install(mp_)
    try:
        from tortoise_tts_amd.api import TextToSpeech
        sds, cfgs = small_setup()
        tts = TextToSpeech(state_dicts=sds, configs=cfgs, max_candidates=16 // world, max_mel_tokens=16, kv_cache=True)
        assert (tts.rank, tts.world) == (rank, world) and tts.split_diffusion == (world >= 2)
        lat = voice_latents(cfgs)
        one = tts.tts(TEXT, conditioning_latents=lat, k=1, **KW)
        if world > 1:  # single winner + conditioning-free guidance: ranks 0 and 1 share the tail, the others take no part in it
            assert getattr(tts.diffusion, "split_steps", None) == (3 if rank < 2 else None)
        best1 = tts.last_best_codes.clone()
        three = tts.tts(TEXT, conditioning_latents=lat, k=3, **KW)
        best3 = tts.last_best_codes.clone()
        if world > 1:  # ONE data-path all_gather per utterance (scores + codes in one packed buffer), two utterances so far
            assert tdist.COLLECTIVE_CALLS.get("all_gather", 0) == 2, tdist.COLLECTIVE_CALLS
        flags = tdist.any_over_ranks([rank == 5, False]) if world > 1 else [True, False]
        assert flags == [True, False]
        if rank == 0:
            assert one is not None and isinstance(three, list) and len(three) == 3
            torch.save({"one": one, "three": three, "best1": best1, "best3": best3}, os.path.join(out_dir, f"tts_w{world}.pt"))
        else:
            assert one is None and three is None
            # every rank selected the same winners from the gathered scores
            torch.save({"best1": best1, "best3": best3}, os.path.join(out_dir, f"best_w{world}_r{rank}.pt"))
    finally:
        mp_.undo()
    if world > 1:
        tdist.barrier()
        dist.destroy_process_group()


def test_tts_control_flow_on_eight_ranks_equals_one_process(tmp_path):
    prev = torch.get_num_threads()  # (the workers run single-threaded: so does the in-process reference, and the setting is put back)
    try:
        _worker(0, 1, 0, str(tmp_path))
    finally:
        torch.set_num_threads(prev)
    mp.spawn(_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    a = torch.load(tmp_path / "tts_w1.pt")
    b = torch.load(tmp_path / "tts_w8.pt")
    assert torch.equal(a["best1"], b["best1"]) and torch.equal(a["best3"], b["best3"]), "sharding over 8 ranks changed the ranked winners"
    assert torch.equal(a["one"], b["one"]), "the single winner's audio differs between 1 and 8 ranks"
    assert all(torch.equal(x, y) for x, y in zip(a["three"], b["three"])), "a k = 3 winner's audio differs between 1 and 8 ranks"
    for r in range(1, 8):
        o = torch.load(tmp_path / f"best_w8_r{r}.pt")
        assert torch.equal(o["best1"], a["best1"]) and torch.equal(o["best3"], a["best3"]), f"rank {r} selected other winners"
