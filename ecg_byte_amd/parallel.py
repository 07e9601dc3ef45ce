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

    def __init__(self, process_group=None, bucket_bytes: int = 25 << 20, single_rank_collectives: bool = False, time_exposed: bool = False,
                 rsag_mb=None, sparse_embedding=None):
        self.pg = process_group
        self.steps = 0             # finish() calls
        self._time_exposed = bool(time_exposed) and torch.cuda.is_available()
        self._exposed = []         # (event before the waits, event after them) per step
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
        # buckets of at least this many bytes leave as reduce-scatter + all-gather (0 = never): ECGB_GRAD_EXCHANGE=rsag (64 MB) or rsag:<MB>
        import os
        mode = os.environ.get("ECGB_GRAD_EXCHANGE", "") if rsag_mb is None else f"rsag:{rsag_mb}"
        self.rsag_bytes = 0
        if mode.startswith("rsag"):
            self.rsag_bytes = int(float(mode.split(":")[1]) * (1 << 20)) if ":" in mode else 64 << 20
        self.rsag_buckets = 0
        # the embedding table's gradient = tied head's dE (dense, final when backward STARTS) + the scatter of the input rows (sparse, final when it
        # ends): with `sparse_embedding` the model sends the dense part first and the sparse part as (ids, rows) -- the exposed tail of a step is
        # then a few MB instead of the 0.54 GB table
        self.sparse_embedding = bool(sparse_embedding) if sparse_embedding is not None else (self.active and self.world > 1)

    def backward_kernels(self):
        """Context manager for one backward pass: one-tile input-gradient GEMMs while an exchange can be in flight."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            if not self.one_tile_backward:
                yield
                return
            from . import decoder_ops as _ops
            before = _ops.get_gemm_backward_persistent()                 # a caller (or a test) may have chosen the one-tile kernels itself: leave what was found
            _ops.set_gemm_backward_persistent(False)
            try:
                yield
            finally:
                _ops.set_gemm_backward_persistent(before)
        return scope()

    def _send(self, t):
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        if self.rsag_bytes and t.numel() * t.element_size() >= self.rsag_bytes and self.world > 1:
            return self._send_rsag(t, op)
        self.pending.append((t, dist.all_reduce(t, op=op, group=self.pg, async_op=True), not self.avg))
        self.collectives += 1

    def _send_rsag(self, t, op):
        """Explicit reduce-scatter + all-gather of one large bucket (2.49 GB of full fine-tune gradients; a reduce-scatter
        followed by an all-gather moves 2 (W-1)/W of the bytes like a ring all-reduce but lets RCCL run each half as a direct exchange over all seven
        xGMI links).  Selected by ECGB_GRAD_EXCHANGE=rsag[:min_mb] or rsag_mb=...; unmeasured on hardware (no multi-GPU node this round): both forms are
        here so that they can be timed the day one exists.  The bucket is cut at a multiple of the world size; the remainder (< world elements) goes
        through a plain all-reduce.  RCCL orders the two collectives on its stream; gloo (CPU tests) has no reduce-scatter: there the shard is cut from
        an all-reduce, which exercises the same bookkeeping."""
        W = self.world
        if not t.is_contiguous():
            raise ValueError("GradAllReduce: a reduce-scatter bucket must be contiguous (a strided view would be reduced into a copy)")
        t = t.view(-1)                                                   # (a parameter's .grad is N-dimensional: cut ELEMENTS, not rows)
        n = (t.numel() // W) * W
        body, rest = t[:n], t[n:]
        r = dist.get_rank(self.pg)
        shard = body.view(W, -1)[r]
        if self.avg:
            w1 = dist.reduce_scatter_tensor(shard, body, op=op, group=self.pg, async_op=True)
            w2 = dist.all_gather_into_tensor(body, shard, group=self.pg, async_op=True)
            self.pending.append((body, w1, False))
            self.pending.append((body, w2, False))
        else:
            w1 = dist.all_reduce(body, op=op, group=self.pg, async_op=True)       # gloo: stands in for the reduce-scatter
            w1.wait()
            body.div_(W)
            full = [torch.empty_like(shard) for _ in range(W)]
            w2 = dist.all_gather(full, shard.contiguous(), group=self.pg, async_op=True)
            w2.wait()
            for k in range(W):
                body.view(W, -1)[k].copy_(full[k])
        self.collectives += 2
        self.rsag_buckets += 1
        if rest.numel():
            self.pending.append((rest, dist.all_reduce(rest, op=op, group=self.pg, async_op=True), not self.avg))
            self.collectives += 1

    def exchange_rows(self, ids, rows):
        """Sparse part of an embedding gradient: every rank contributes (ids [n] int64, unique and sorted; rows [n, H]) -- the per-id sums of its own
        batch -- and gets back the list [(ids_r, rows_r)] of all ranks in rank order (its own included).  One all-gather of the counts (host
        sync: a few bytes), then padded all-gathers of the ids and the rows: W x max_n x H elements instead of the whole table (0.54 GB for
        Llama-3.2-1B's 132 k x 2048 embedding; a batch touches a few thousand distinct ids).  The caller adds rows_r / W to its table in rank order."""
        if not self.active or self.world == 1:
            return [(ids, rows)]
        W = self.world
        n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
        counts = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(counts, n, group=self.pg)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        pid = torch.full((cap,), -1, dtype=torch.int64, device=ids.device)
        pid[: ids.numel()] = ids
        prow = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        prow[: rows.shape[0]] = rows
        gid = [torch.empty_like(pid) for _ in range(W)]
        grow = [torch.empty_like(prow) for _ in range(W)]
        dist.all_gather(gid, pid, group=self.pg)
        dist.all_gather(grow, prow, group=self.pg)
        self.collectives += 3
        return [(gid[k][: counts[k]], grow[k][: counts[k]]) for k in range(W)]

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
        self.steps += 1
        ev = None
        if self._time_exposed and self.active:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for t, work, divide in self.pending:
            work.wait()
            if divide and self.world > 1:
                t.div_(self.world)
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        self.pending = []

    def exposed_ms(self) -> float:
        """Total time the compute stream spent waiting in finish() for buckets still in flight (needs time_exposed=True; synchronises)."""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self._exposed))
