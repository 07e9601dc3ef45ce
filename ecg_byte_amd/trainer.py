"""Host side of the HIP BPE trainer (reference: rust_bpe.byte_pair_encoding,
ecg_byte/rust_bpe/src/lib.rs:58-125).  torch provides device buffers; the work is
`ecgb_bpe_train_hip`."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .tokenizer import _ptr, _stream_ptr


def set_train_grid(workgroups: int):
    """Tests and tuning: workgroups of the trainer's count and rewrite passes (0 = default); see ecgb_set_bpe_train_grid in include/ecgbyte.h."""
    _lib.check(_lib.lib().ecgb_set_bpe_train_grid(int(workgroups)))


def set_train_form(form: int):
    """Tests and tuning: 0 = slotted ranges, one pass per merge (default); 1 = the same with 32-bit ids; 2 = round 4's two passes; see ecgb_set_bpe_train_form."""
    _lib.check(_lib.lib().ecgb_set_bpe_train_form(int(form)))


def set_train_fused(on: bool = True):
    """Tests and tuning (forms 0 / 1): the next merge's row maxima inside the merge's launch (one launch per merge; measured 2 % slower than the default, a launch of their
    own -- EXPERIMENTS.md R6); see ecgb_set_bpe_train_fused.  set_train_fused(False) restores the default."""
    _lib.check(_lib.lib().ecgb_set_bpe_train_fused(int(bool(on))))


def bpe_train_device(text: torch.Tensor, num_merges: int):
    """text: CUDA uint8 1-D tensor.  Returns device tensors (ids int32[n], n_ids int64[1],
    pairs int32[num_merges, 2], n_done int32[1]); nothing is synchronised."""
    if not (isinstance(text, torch.Tensor) and text.is_cuda and text.dtype == torch.uint8):
        raise TypeError("bpe_train_device needs a CUDA uint8 tensor (no CPU fallback)")
    text = text.contiguous().view(-1)
    n = text.numel()
    dev = text.device
    L = _lib.lib()
    nb = L.ecgb_bpe_train_scratch_bytes(n, num_merges)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    pairs = torch.zeros((max(1, num_merges), 2), dtype=torch.int32, device=dev)
    n_done = torch.zeros(1, dtype=torch.int32, device=dev)
    ids = torch.empty(max(1, n), dtype=torch.int32, device=dev)
    n_ids = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(L.ecgb_bpe_train_hip(_ptr(text), n, int(num_merges), _ptr(pairs), _ptr(n_done), _ptr(ids),
                                    _ptr(n_ids), _ptr(scratch), nb, _stream_ptr()))
    return ids, n_ids, pairs, n_done


def bpe_train(text_bytes, num_merges: int):
    """numpy uint8 array / bytes -> (ids list[int], pairs list[(left, right)])."""
    raw = np.frombuffer(bytes(text_bytes), dtype=np.uint8) if not isinstance(text_bytes, np.ndarray) else text_bytes
    t = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.uint8).copy()).cuda()
    ids, n_ids, pairs, n_done = bpe_train_device(t, num_merges)
    k = int(n_done.item())
    m = int(n_ids.item())
    return ids[:m].cpu().tolist(), [tuple(p) for p in pairs[:k].cpu().tolist()]


# ---- corpus sharded over ranks ------------------------------------------------------------------------
class HipShard:
    """One rank's slice of the corpus on its GPU: the step-wise C ABI of include/ecgbyte.h (ecgb_bpe_shard_*)."""

    def __init__(self, text: torch.Tensor, num_merges: int):
        if not (isinstance(text, torch.Tensor) and text.is_cuda and text.dtype == torch.uint8):
            raise TypeError("HipShard needs a CUDA uint8 tensor (no CPU fallback)")
        _lib.require_current(text.device)
        self.text = text.contiguous().view(-1)
        self.n, self.num_merges, self.dev = self.text.numel(), int(num_merges), text.device
        L = _lib.lib()
        nb = L.ecgb_bpe_train_scratch_bytes(self.n, self.num_merges)
        self.scratch = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        self.h = L.ecgb_bpe_shard_create(self.n, self.num_merges, _ptr(self.scratch), nb)
        if not self.h:
            _lib.check(-1)
        self._views = {}

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            _lib.lib().ecgb_bpe_shard_destroy(h)
            self.h = None

    def _view(self, which):
        """The table / slab inside the scratch buffer as an int64 tensor (what the collectives reduce)."""
        if which not in self._views:
            import ctypes as C
            n = C.c_size_t()
            ptr = getattr(_lib.lib(), "ecgb_bpe_shard_" + which)(self.h, C.byref(n))
            off = ptr - self.scratch.data_ptr()
            self._views[which] = self.scratch[off: off + 8 * n.value].view(torch.int64)
        return self._views[which]

    def new_words(self, n):
        return torch.zeros(n, dtype=torch.int64, device=self.dev)

    def begin(self, summary):
        _lib.check(_lib.lib().ecgb_bpe_shard_begin(self.h, _ptr(self.text), _ptr(summary), _stream_ptr()))

    def count(self, gathered, rank, world):
        _lib.check(_lib.lib().ecgb_bpe_shard_count(self.h, _ptr(gathered), rank, world, _stream_ptr()))

    def table(self):
        return self._view("table")

    def pick(self, i, summary):
        _lib.check(_lib.lib().ecgb_bpe_shard_pick(self.h, i, _ptr(summary), _stream_ptr()))

    def merge(self, i, gathered, rank, world):
        _lib.check(_lib.lib().ecgb_bpe_shard_merge(self.h, i, _ptr(gathered), rank, world, _stream_ptr()))

    def slab(self):
        return self._view("slab")

    def apply(self):
        _lib.check(_lib.lib().ecgb_bpe_shard_apply(self.h, _stream_ptr()))

    def finish(self):
        pairs = torch.zeros((max(1, self.num_merges), 2), dtype=torch.int32, device=self.dev)
        n_done = torch.zeros(1, dtype=torch.int32, device=self.dev)
        ids = torch.empty(max(1, self.n), dtype=torch.int32, device=self.dev)
        n_ids = torch.zeros(1, dtype=torch.int64, device=self.dev)
        _lib.check(_lib.lib().ecgb_bpe_shard_finish(self.h, _ptr(pairs), _ptr(n_done), _ptr(ids), _ptr(n_ids), _stream_ptr()))
        k, m = int(n_done.item()), int(n_ids.item())
        return ids[:m].cpu().tolist(), [tuple(p) for p in pairs[:k].cpu().tolist()]


def _reduce_(t, group):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) != "nccl":       # gloo (the CPU / one-GPU tests): through host memory
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def _gather_(out, t, group):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) != "nccl":
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, t.cpu(), group=group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, t, group=group)


def bpe_train_sharded(shard, num_merges: int, group=None):
    """byte_pair_encoding (lib.rs:58-125) on a corpus split into contiguous slices, one per rank of `group` (RCCL over xGMI with
    backend "nccl").  `shard` holds this rank's slice and does the local work of each step (HipShard on a GPU; the CPU tests drive
    the same protocol with oracle/sharded_trainer.py); this function is the exchange: per merge one all-gather of 8 words per rank
    and one all-reduce of 6 x V words.  Returns (this rank's final ids, the merges' (left, right) pairs -- identical on every rank)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    summary, gathered = shard.new_words(8), shard.new_words(8 * world)
    shard.begin(summary)
    _gather_(gathered, summary, group)
    shard.count(gathered, rank, world)
    _reduce_(shard.table(), group)
    for i in range(int(num_merges)):
        shard.pick(i, summary)
        _gather_(gathered, summary, group)
        shard.merge(i, gathered, rank, world)
        _reduce_(shard.slab(), group)
        shard.apply()
    return shard.finish()
