"""Candidate sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU path: it loops `num_autoregressive_samples // batch` batches on one
device (tortoise/api.py:407-427) and ranks them with CLVP (api.py:447-477).  Candidates are i.i.d.
given (voice latent, text), so rank r decodes and scores N/R of them with replicated weights and a
single all_gather of (scores, codes) lets every rank compute the identical top-k.
Payload: N/R f32 scores + N/R x 500 int32 codes per rank (64 KB at 32 candidates) — latency-bound,
one collective per utterance; nothing else on the path communicates.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_index=None):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process).
    Backend: nccl (== RCCL on ROCm) when a GPU is present, gloo otherwise (CPU tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TT_DIST_SHARE_DEVICE=1 is a TEST mode for boxes with fewer GPUs than ranks: ranks share devices and talk over
    # gloo (RCCL refuses two ranks on one device).  It exercises the whole multi-rank control flow - sharding, the
    # gather, winner selection, rank-0 rendering, timing reduction - not the xGMI transport.
    share = os.environ.get("TT_DIST_SHARE_DEVICE") == "1"
    if torch.cuda.is_available():
        if device_index is None:
            device_index = local % torch.cuda.device_count() if share else local
        torch.cuda.set_device(device_index)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() and not share else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _host_staged():
    """gloo moves host memory: device tensors are staged through the CPU (CPU tests and the shared-device test mode)."""
    return dist.get_backend() == "gloo"


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def shard_range(n, rank, world_size):
    """Contiguous candidate range of `rank`; requires n % world_size == 0 so the all_gather is regular."""
    if n % world_size != 0:
        raise ValueError(f"num_autoregressive_samples={n} must be divisible by the number of GPUs ({world_size})")
    per = n // world_size
    return rank * per, (rank + 1) * per


def gather_candidates(scores_local, codes_local):
    """all_gather of CLVP scores f32 [n] and codes int32 [n, M] -> global ([N], [N, M]) on every rank,
    ordered by global candidate index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return scores_local, codes_local
    ws = dist.get_world_size()
    scores_local = scores_local.contiguous()
    codes_local = codes_local.to(torch.int32).contiguous()
    dev = scores_local.device
    if _host_staged() and dev.type != "cpu":
        scores_local, codes_local = scores_local.cpu(), codes_local.cpu()
    s_all = torch.empty(ws * scores_local.shape[0], dtype=scores_local.dtype, device=scores_local.device)
    c_all = torch.empty(ws * codes_local.shape[0], codes_local.shape[1], dtype=torch.int32, device=codes_local.device)
    dist.all_gather_into_tensor(s_all, scores_local)
    dist.all_gather_into_tensor(c_all, codes_local)
    return s_all.to(dev), c_all.to(dev)


def topk_lowest_index(scores, k):
    """torch.topk(scores, k).indices (api.py:477) with ties broken towards the lowest global index,
    so every rank — and the single-GPU run — selects the same candidates."""
    order = torch.sort(-scores.double(), stable=True).indices
    return order[:k]


def barrier():
    if dist.is_initialized():
        dist.barrier()


_PAIR = None


def pair_group():
    """Process group {0, 1} for the split diffusion tail (created once; new_group is collective over ALL ranks)."""
    global _PAIR
    if _PAIR is None and dist.is_initialized() and dist.get_world_size() >= 2:
        _PAIR = dist.new_group([0, 1])
    return _PAIR


def exchange_rows(rows, mine):
    """rows[r] <- participant r's `mine` over the pair group: the per-step exchange of the split diffusion tail
    (f32 [S][200] per rank, ~0.7 MB at S = 870; xGMI point-to-point between GPU 0 and GPU 1)."""
    group = pair_group()
    if _host_staged() and mine.device.type != "cpu":
        host = torch.empty(rows.shape, dtype=rows.dtype)
        dist.all_gather_into_tensor(host.view(-1), mine.cpu().view(-1), group=group)
        rows.copy_(host)
    else:
        dist.all_gather_into_tensor(rows.view(-1), mine.contiguous().view(-1), group=group)


def broadcast_int(value, src=0):
    """`value` of rank `src` on every rank (host integer; e.g. the utterance seed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    dev = "cpu" if _host_staged() or not torch.cuda.is_available() else "cuda"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


def max_over_ranks(value):
    """MAX of a host scalar over all ranks (bench.py's timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = "cpu" if _host_staged() or not torch.cuda.is_available() else "cuda"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
