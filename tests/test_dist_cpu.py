"""World-size-2 `gloo` test of the candidate-sharding exchange (tortoise_tts_amd/dist.py): each rank holds
N/R (score, codes) rows; after the single all_gather every rank must select the same top-k as one
process holding all N rows, with ties resolved to the lowest global index."""
import os
import socket

import numpy as np
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


def _worker(rank, world, port, N, M, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    tdist.init_from_env()
    assert tdist.world() == (rank, world)
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(N, generator=g)
    scores[5] = scores[11]  # a tie across the shard boundary region
    scores[N - 1] = scores[0] = scores.max() + 1  # tie for first place between rank 0 and the last rank
    codes = torch.randint(0, 8192, (N, M), generator=g, dtype=torch.int32)
    lo, hi = tdist.shard_range(N, rank, world)
    s_all, c_all = tdist.gather_candidates(scores[lo:hi].clone(), codes[lo:hi].clone())
    assert torch.equal(s_all, scores) and torch.equal(c_all, codes)
    best = tdist.topk_lowest_index(s_all, k)
    np.save(os.path.join(out_dir, f"best_{rank}.npy"), best.numpy())
    np.save(os.path.join(out_dir, f"codes_{rank}.npy"), c_all[best].numpy())
    tdist.barrier()
    dist.destroy_process_group()


def test_sharded_topk_matches_single_process(tmp_path):
    N, M, k, world = 16, 20, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), N, M, k, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(N, generator=g)
    scores[5] = scores[11]
    scores[N - 1] = scores[0] = scores.max() + 1
    want = tdist.topk_lowest_index(scores, k).numpy()
    assert want[0] == 0 and want[1] == N - 1  # equal scores -> lowest global index first
    b0, b1 = np.load(tmp_path / "best_0.npy"), np.load(tmp_path / "best_1.npy")
    assert np.array_equal(b0, want) and np.array_equal(b1, want)
    assert np.array_equal(np.load(tmp_path / "codes_0.npy"), np.load(tmp_path / "codes_1.npy"))


def test_shard_range_and_topk_rules():
    assert tdist.shard_range(256, 3, 8) == (96, 128)
    try:
        tdist.shard_range(10, 0, 4)
        raise AssertionError("uneven shard accepted")
    except ValueError:
        pass
    s = torch.tensor([1.0, 3.0, 3.0, 2.0])
    assert tdist.topk_lowest_index(s, 2).tolist() == [1, 2]


def _split_worker(rank, world, port, out_dir):
    """The split diffusion tail's host protocol with a stand-in denoiser: rank r < 2 produces row r of every step,
    the pair exchanges rows, both apply the same update -> identical state on ranks 0 and 1; rank 2 takes no part."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    tdist.init_from_env()
    tdist._PAIR = None
    assert tdist.pair_group() is not None  # collective: every rank, including non-members, creates it
    seed = tdist.broadcast_int(1234 + rank)  # rank 0's value wins
    assert seed == 1234
    assert tdist.max_over_ranks(float(rank)) == float(world - 1)
    if rank < 2:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(7, 5, generator=g)
        rows = torch.empty(2, 7, 5)
        for step in range(4):
            mine = x * (rank + 1) + step  # "model row" of this participant
            tdist.exchange_rows(rows, mine)
            assert torch.equal(rows[rank], mine)
            x = 0.5 * x + 0.25 * (rows[0] - rows[1])
        np.save(os.path.join(out_dir, f"x_{rank}.npy"), x.numpy())
    tdist.barrier()
    dist.destroy_process_group()


def test_split_tail_protocol_keeps_pair_in_lockstep(tmp_path):
    world = 3
    mp.spawn(_split_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    x0, x1 = np.load(tmp_path / "x_0.npy"), np.load(tmp_path / "x_1.npy")
    assert np.array_equal(x0, x1)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(7, 5, generator=g)
    for step in range(4):
        x = 0.5 * x + 0.25 * ((x * 1 + step) - (x * 2 + step))
    assert np.array_equal(x0, x.numpy())



def _collect_worker(rank, world, port, out_dir):
    """k = 3 winners over 2 ranks: winner i is rendered on rank i % 2; rank 0 must end up with all three waveforms (two
    of its own + one point-to-point receive), rank 1 with None."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    tdist.init_from_env()
    k = 3
    mine = {i: torch.full((1, 1, 10 + i), float(i)) for i in range(k) if i % world == rank}
    got = tdist.collect_on_rank0(mine, k)
    if rank == 0:
        assert sorted(got) == [0, 1, 2]
        for i in range(k):
            assert got[i].shape == (1, 1, 10 + i) and bool((got[i] == float(i)).all())
    else:
        assert got is None
    # k = 1: the winner is on rank 0, nothing is sent
    got = tdist.collect_on_rank0({0: torch.zeros(1, 1, 4)} if rank == 0 else {}, 1)
    assert (got is not None) == (rank == 0)
    tdist.barrier()
    dist.destroy_process_group()


def test_rendered_winners_are_collected_on_rank0_only():
    mp.spawn(_collect_worker, args=(2, _free_port(), ""), nprocs=2, join=True)


def _riding_hood_chunks():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_split.json")) as f:
        chunks = json.load(f)[2]["chunks"]  # the reference's own 15 chunks of tortoise/data/riding_hood.txt
    # pre-tokenised stand-ins (the BPE vocabulary is a reference data file that is not on every box): one id per character
    return [[1 + (ord(c) % 250) for c in ch][:24] for ch in chunks]


def _longform_worker(rank, world, port, out_dir, ubatch=1, nsamp=2, tag=""):
    """BASELINE config #4 on CPU stand-ins: chunk j is rendered by rank j % world with the complete (unsharded) pipeline;
    ubatch > 1: a rank renders its chunks in shared batches (TextToSpeech.tts_many)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if world > 1:
        tdist.init_from_env()
    import pytest
    from tests import fake_stages
    from tests.test_api_flow_cpu import small_setup, voice_latents
    mp_ = pytest.MonkeyPatch()
    fake_stages.install(mp_)
    try:
        from tortoise_tts_amd.api import TextToSpeech
        from tortoise_tts_amd.longform import read_long_form
        sds, cfgs = small_setup()
        tts = TextToSpeech(state_dicts=sds, configs=cfgs, max_candidates=8, max_mel_tokens=16, kv_cache=True, candidate_sharding=False,
                           utterance_batch=ubatch)
        calls = []
        orig = tts.tts_with_preset
        tts.tts_with_preset = lambda *a, **k: (calls.append(k.get("use_deterministic_seed")), orig(*a, **k))[1]
        full, clips = read_long_form(tts, _riding_hood_chunks(), preset="ultra_fast", conditioning_latents=voice_latents(cfgs), seed=5,
                                     texts_are_chunks=True, num_autoregressive_samples=nsamp, diffusion_iterations=2, max_mel_tokens=10)
        mine = len([j for j in range(15) if j % world == rank])
        if ubatch > 1 and mine > 1:
            assert not calls and tts.ar.group_batches >= 1  # the chunks went through shared decode batches, not one call per chunk
            assert sum(tts.diffusion.batched) + (mine - sum(tts.diffusion.batched)) == mine
            # round 6: the CLVP ranking of a wave is ONE grouped call (one speech-tower pass), not one call per chunk
            assert sum(tts.clvp.grouped) == mine and len(tts.clvp.grouped) == -(-mine // ubatch), tts.clvp.grouped
        else:
            assert calls and all(c == 5 for c in calls)  # the same seed for every chunk (read.py:54, 70-71)
            assert len(calls) == mine
        if rank == 0:
            assert len(clips) == 15 and full.shape[-1] == sum(c.shape[-1] for c in clips)
            torch.save((full, clips), os.path.join(out_dir, f"longform_w{world}{tag}.pt"))
        else:
            assert full is None and clips is None
    finally:
        mp_.undo()
    if world > 1:
        tdist.barrier()
        dist.destroy_process_group()


def test_longform_chunks_spread_over_ranks_match_sequential(tmp_path):
    _longform_worker(0, 1, 0, str(tmp_path))                                            # sequential, like read.py
    mp.spawn(_longform_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)  # chunk j -> rank j % 2
    f1, c1 = torch.load(tmp_path / "longform_w1.pt")
    f2, c2 = torch.load(tmp_path / "longform_w2.pt")
    assert torch.equal(f1, f2), "spreading the chunks over two ranks changed the audio"
    assert all(torch.equal(a, b) for a, b in zip(c1, c2))  # chunk order preserved
    assert len({tuple(c.flatten()[:64].tolist()) for c in c1}) > 1  # different chunks give different audio


def test_longform_shared_batches_match_one_chunk_after_the_other(tmp_path):
    """A rank that renders its chunks 3 at a time (shared decode batches + shared denoiser passes, tts_many) inside a 2-rank job
    returns the audio of the sequential single-process reading (same seed, same candidate count)."""
    _longform_worker(0, 1, 0, str(tmp_path), 1, 4, "_seq")
    mp.spawn(_longform_worker, args=(2, _free_port(), str(tmp_path), 3, 4, "_ub3"), nprocs=2, join=True)
    f1, c1 = torch.load(tmp_path / "longform_w1_seq.pt")
    f2, c2 = torch.load(tmp_path / "longform_w2_ub3.pt")
    assert torch.equal(f1, f2) and all(torch.equal(a, b) for a, b in zip(c1, c2))
