"""Multi-GPU glue for the encode path.  Records are independent, so the batch dimension is
sharded across ranks with the reference's `DistributedSampler` index rule
(ecg_byte/main.py:239-243) and NO collective runs on the data path; the only collectives are the
bookkeeping reductions below (RCCL on GPUs -- backend "nccl" -- or gloo in the CPU tests)."""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def shard_indices(n_records: int, rank: int, world: int, shuffle: bool = False, seed: int = 0, epoch: int = 0) -> list[int]:
    """Indices rank `rank` processes: torch's DistributedSampler(drop_last=False) rule (the reference's sampler,
    ecg_byte/main.py:239-243) -- optionally a permutation seeded with seed + epoch, padded by wrapping to a
    multiple of `world`, then strided."""
    if n_records == 0:
        return []
    per = math.ceil(n_records / world)
    total = per * world
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_records, generator=g).tolist()
    else:
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


class GradAllReduce:
    """Data-parallel gradient exchange for HipCausalLM (the reference wraps its model in
    DistributedDataParallel, ecg_byte/main.py:165: bucketed all-reduce(avg) of trainable grads,
    fired during backward).  Here the model calls `on_grads_ready(params)` as soon as a layer's
    gradients are final; each call enqueues one asynchronous all-reduce per tensor (RCCL over
    xGMI with backend "nccl"; gloo in the CPU test), overlapping the exchange of layer i with the
    backward of layer i-1.  `finish()` waits for all of them and turns sums into means."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.pending = []
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    def on_grads_ready(self, params):
        if self.world == 1:
            return
        for p in params:
            if p.grad is not None:
                self.pending.append((p, dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))

    def finish(self):
        for p, work in self.pending:
            work.wait()
            p.grad.div_(self.world)
        self.pending = []
