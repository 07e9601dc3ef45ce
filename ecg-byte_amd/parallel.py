"""Multi-GPU glue for the encode path.  Records are independent, so the batch dimension is
sharded across ranks with the reference's `DistributedSampler` index rule
(ecg_byte/main.py:239-243) and NO collective runs on the data path; the only collectives are the
bookkeeping reductions below (RCCL on GPUs -- backend "nccl" -- or gloo in the CPU tests)."""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def shard_indices(n_records: int, rank: int, world: int) -> list[int]:
    """Indices rank `rank` processes: torch's DistributedSampler(shuffle=False, drop_last=False)
    rule -- pad by wrapping to a multiple of `world`, then stride."""
    if n_records == 0:
        return []
    per = math.ceil(n_records / world)
    total = per * world
    idx = list(range(n_records))
    pad = total - n_records
    if pad:
        idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx[rank:total:world]


def reduce_step_stats(wall_s: float, tokens: int, device) -> tuple[float, int]:
    """MAX of the per-rank wall time and SUM of the per-rank token counts."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return wall_s, tokens
    t = torch.tensor([wall_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    k = torch.tensor([tokens], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.SUM)
    return float(t.item()), int(k.item())


def gather_counts(counts: torch.Tensor) -> torch.Tensor:
    """All ranks' per-record token counts, rank-major (for ragged bookkeeping on rank 0)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return counts
    out = [torch.empty_like(counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, counts)
    return torch.cat(out)
