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
    DistributedDataParallel, ecg_byte/main.py:165: all-reduce(avg) of trainable grads in 25 MB buckets, fired
    during backward).

    HipCausalLM keeps every trainable gradient in ONE flat bf16 buffer laid out in the order backward finishes
    them (last layer first) and calls `on_flat_ready(flat, lo, hi)` when the gradients of elements [lo, hi) are
    final.  Adjacent ranges are merged until a bucket holds `bucket_bytes` (25 MB, DDP's default), then ONE
    asynchronous all-reduce goes out for the whole range -- a full fine-tune layer of Llama-3.2-1B is 121 MB, i.e.
    one collective per layer overlapping the backward of the layers below it; all LoRA adapters of the model
    (a few MB per layer) leave in three or four.  With RCCL (backend "nccl") the reduction is ncclAvg: the mean
    comes out of the collective itself, nothing is divided afterwards.  gloo (CPU tests) has no AVG: SUM, then one
    in-place scale per bucket.  `on_grads_ready(params)` is the per-tensor path for gradients that are not part of
    a flat buffer.  `finish()` sends what is left and waits for everything."""

    def __init__(self, process_group=None, bucket_bytes: int = 25 << 20, single_rank_collectives: bool = False):
        self.pg = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.pending = []          # (tensor, work, needs_divide)
        self.open = None           # (flat, lo, hi): ready, not yet sent
        self.collectives = 0       # all-reduces issued since construction (tests)
        ok = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if ok else 1
        self.avg = ok and dist.get_backend(process_group) == "nccl"
        # world size 1 needs no exchange; `single_rank_collectives` issues the collectives anyway (the one-GPU box's test of the
        # RCCL path: communicator, async all-reduce, stream ordering against the ctypes-launched kernels)
        self.active = ok and (self.world > 1 or single_rank_collectives)
        # The collectives run beside the backward's input-gradient GEMMs and hold CUs: persistent workgroups with a static share of the
        # tiles would wait for the late starters; one tile per workgroup degrades gracefully (include/ecgbyte_decoder.h).  The switch is
        # process-global in the C library, so it is NOT set here: the model's backward turns it off for the duration of a backward pass
        # with an active exchange and restores it (`backward_kernels()`), and later models / evaluation in the process keep the
        # persistent kernels.  The one-rank RCCL leg (`single_rank_collectives`) makes the same choice as world > 1.
        self.one_tile_backward = bool(self.active and torch.cuda.is_available())

    def backward_kernels(self):
        """Context manager for one backward pass: one-tile input-gradient GEMMs while an exchange can be in flight."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            if not self.one_tile_backward:
                yield
                return
            from . import decoder_ops as _ops
            _ops.set_gemm_backward_persistent(False)
            try:
                yield
            finally:
                _ops.set_gemm_backward_persistent(True)
        return scope()

    def _send(self, t):
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.pending.append((t, dist.all_reduce(t, op=op, group=self.pg, async_op=True), not self.avg))
        self.collectives += 1

    def _flush(self):
        if self.open is not None:
            flat, lo, hi = self.open
            self.open = None
            self._send(flat[lo:hi])

    def on_flat_ready(self, flat, lo, hi):
        if not self.active or hi <= lo:
            return
        if self.open is not None and self.open[0] is flat and self.open[2] == lo:
            self.open = (flat, self.open[1], hi)
        else:
            self._flush()
            self.open = (flat, lo, hi)
        if (self.open[2] - self.open[1]) * flat.element_size() >= self.bucket_bytes:
            self._flush()

    def on_grads_ready(self, params):
        if not self.active:
            return
        self._flush()
        for p in params:
            if p.grad is not None:
                self._send(p.grad)

    def finish(self):
        self._flush()
        for t, work, divide in self.pending:
            work.wait()
            if divide and self.world > 1:
                t.div_(self.world)
        self.pending = []
