"""Host side of the HIP BPE trainer (reference: rust_bpe.byte_pair_encoding,
ecg_byte/rust_bpe/src/lib.rs:58-125).  torch provides device buffers; the work is
`ecgb_bpe_train_hip`."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .tokenizer import _ptr, _stream_ptr


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
