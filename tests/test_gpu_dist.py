"""-m gpu: the RCCL (backend "nccl") branches of tortoise_tts_amd/dist.py on real device tensors.

The GPU boxes of this project expose ONE GPU, so the multi-rank flow itself is covered by the gloo tests
(tests/test_dist_cpu.py, world_size 2 and 3).  What those cannot reach is the transport branch every collective takes on a GPU
node: device-resident payloads handed to RCCL without host staging.  Here a world_size-1 `nccl` process group is created on
the one GPU and dist.FORCE_COLLECTIVES makes every helper run its collective anyway (a one-rank all_gather / broadcast /
all_reduce is still a real RCCL kernel launch on the communicator), so `gather_candidates`, `exchange_rows`, `broadcast_int`,
`max_over_ranks`, `barrier` and the candidate pipeline built on them (`api.TextToSpeech` sharding arithmetic) execute the code
path an 8-GPU node runs, with results checked against the identity they must reduce to at one rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from tortoise_tts_amd import dist as tdist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def rccl_world1():
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    assert not dist.is_initialized()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    tdist.FORCE_COLLECTIVES = True
    tdist._PAIR = None
    try:
        yield
    finally:
        tdist.FORCE_COLLECTIVES = False
        tdist._PAIR = None
        dist.destroy_process_group()


def test_rccl_branches_of_every_collective(rccl_world1):
    assert dist.get_backend() == "nccl" and not tdist._host_staged()
    g = torch.Generator().manual_seed(0)
    # gather of (CLVP scores, codes as int16 words): odd code count per rank exercises the padding word
    scores = torch.randn(32, generator=g).cuda()
    codes = torch.randint(0, 8194, (32, 201), generator=g, dtype=torch.int32).cuda()
    s_all, c_all = tdist.gather_candidates(scores, codes)
    assert s_all.is_cuda and c_all.is_cuda and c_all.dtype == torch.int32
    assert torch.equal(s_all, scores) and torch.equal(c_all, codes)
    best = tdist.topk_lowest_index(s_all, 3)
    assert torch.equal(best.cpu(), torch.sort(-scores.double().cpu(), stable=True).indices[:3])
    # per-step exchange of the split diffusion tail (f32 [S][200] per participant) over the pair group
    mine = torch.randn(870, 200, generator=g).cuda()
    rows = torch.zeros(1, 870, 200, device="cuda")
    tdist.exchange_rows(rows, mine)
    assert torch.equal(rows[0], mine)
    # utterance seed broadcast and the bench's max-over-ranks reduction travel as device tensors on this backend
    assert tdist.broadcast_int(1234567891234) == 1234567891234
    assert tdist.max_over_ranks(0.71875) == 0.71875
    tdist.barrier()
    # rendered clips: with one rank every winner is already on rank 0 and nothing is sent
    wavs = {0: torch.zeros(1, 1, 8)}
    assert tdist.collect_on_rank0(wavs, 1) is wavs
    with pytest.raises(ValueError):
        tdist.gather_candidates(scores, codes + 40000)  # does not fit the int16 payload


def test_shard_ranges_cover_all_candidates():
    for n, ws in ((256, 8), (96, 4), (16, 2)):
        spans = [tdist.shard_range(n, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        tdist.shard_range(100, 0, 8)
