"""Device tokenizer handle: the trie the reference rebuilds inside every
`rust_bpe.encode_text` call (ecg_byte/rust_bpe/src/lib.rs:153-161) is built once here and kept
in HBM; batches of ECG records are quantised and encoded on the GPU.

torch is used for device memory and streams only; all compute goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def flatten_merges(merges):
    """[(list[int] expansion, int id), ...] -> (flat u32, offsets u32[n+1], ids u32[n])."""
    n = len(merges)
    lens = np.fromiter((len(m[0]) for m in merges), dtype=np.int64, count=n)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(lens, out=offsets[1:])
    flat = np.empty(int(offsets[-1]), dtype=np.uint32)
    ids = np.empty(n, dtype=np.uint32)
    for i, (seq, tid) in enumerate(merges):
        flat[offsets[i]:offsets[i + 1]] = seq
        ids[i] = tid
    return flat, offsets, ids


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_encode_plan(mode: int) -> None:
    """0 = automatic, 1 = workgroup-per-stream kernel, 2 / 3 = wave-per-stream kernel, 4 = lane-per-chunk kernel (encode_long_kernel; opt-in), 6 = the same with stager and walker waves in one workgroup (encode_pipe_kernel) (tests/tuning)."""
    _lib.check(_lib.lib().ecgb_set_encode_plan(int(mode)))


class HipTokenizer:
    """Immutable device tokenizer built from the reference's `merges` list."""

    def __init__(self, merges):
        L = _lib.lib()
        flat, offsets, ids = flatten_merges(merges)
        if torch.cuda.is_available():
            torch.cuda.current_device()  # make torch's device/context current before the upload
            torch.zeros(1, device="cuda")
        h = C.c_void_p()
        u32p = C.POINTER(C.c_uint32)
        _lib.check(L.ecgb_tokenizer_create(flat.ctypes.data_as(u32p), offsets.ctypes.data_as(u32p),
                                           ids.ctypes.data_as(u32p), len(merges), C.byref(h)))
        self._h = h
        self.n_merges = len(merges)
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(L.ecgb_tokenizer_info(h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_nodes, self.max_depth, self.n_classes = a.value, b.value, c.value
        self._scratch = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ecgb_tokenizer_destroy(h)
            except Exception:
                pass
            self._h = None

    def nodes(self) -> np.ndarray:
        """Packed trie nodes (host copy): bitmap | first_child<<32 | token<<48."""
        out = np.empty(self.n_nodes, dtype=np.uint64)
        _lib.lib().ecgb_tokenizer_copy_nodes(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size)
        return out

    # ---- device paths -------------------------------------------------------------------
    def _get_scratch(self, batch: int, n: int, device) -> torch.Tensor:
        need = _lib.lib().ecgb_encode_scratch_bytes(self._h, batch, n)
        key = (device.index, torch.cuda.current_stream().cuda_stream)
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(need, dtype=torch.uint8, device=device)
            self._scratch[key] = buf
        return buf

    def quantize_encode(self, signal: torch.Tensor, percentiles, ids_stride: int | None = None,
                        out: tuple | None = None):
        """Quantise + encode a batch of records (reference: data_loader.py:74-76 per sample).

        signal: CUDA float64 tensor `(B, 12, L)` or `(B, n)`, C-contiguous.
        Returns `(ids, counts)`: int32 `(B, ids_stride)` and int32 `(B,)`; record b's tokens are
        `ids[b, :counts[b]]` (entries past `counts[b]` are unspecified).  `ids_stride`
        defaults to the worst case n (one token per symbol); tokens past it are dropped.
        """
        if not (isinstance(signal, torch.Tensor) and signal.is_cuda):
            raise TypeError("quantize_encode needs a CUDA tensor (no CPU fallback)")
        if signal.dtype != torch.float64:
            raise TypeError("signal must be float64 (the reference quantises in float64)")
        _lib.require_current(signal.device)
        if not signal.is_contiguous():
            signal = signal.contiguous()
        B = signal.shape[0]
        n = signal.numel() // B if B else 0
        stride = int(ids_stride) if ids_stride else max(1, n)
        if out is None:
            ids = torch.empty((B, stride), dtype=torch.int32, device=signal.device)
            counts = torch.empty((B,), dtype=torch.int32, device=signal.device)
        else:
            ids, counts = out
        if B == 0:
            return ids, counts
        scratch = self._get_scratch(B, n, signal.device)
        _lib.check(_lib.lib().ecgb_quantize_encode_hip(
            self._h, _ptr(signal), B, n, float(percentiles["percentile_1"]),
            float(percentiles["percentile_99"]), _ptr(ids), stride, _ptr(counts), _ptr(scratch),
            scratch.numel(), _stream_ptr()))
        return ids, counts

    def encode_bytes(self, text: torch.Tensor, ids_stride: int | None = None):
        """Encode a batch of raw byte streams: CUDA uint8 `(B, n)` -> `(ids, counts)`."""
        if not (isinstance(text, torch.Tensor) and text.is_cuda and text.dtype == torch.uint8):
            raise TypeError("encode_bytes needs a CUDA uint8 tensor (no CPU fallback)")
        _lib.require_current(text.device)
        if text.dim() == 1:
            text = text[None]
        text = text.contiguous()
        B, n = text.shape
        stride = int(ids_stride) if ids_stride else max(1, n)
        ids = torch.empty((B, stride), dtype=torch.int32, device=text.device)
        counts = torch.empty((B,), dtype=torch.int32, device=text.device)
        if B == 0:
            return ids, counts
        scratch = self._get_scratch(B, n, text.device)
        _lib.check(_lib.lib().ecgb_encode_hip(self._h, _ptr(text), B, n, _ptr(ids), stride, _ptr(counts),
                                              _ptr(scratch), scratch.numel(), _stream_ptr()))
        return ids, counts


def quantize(signal: torch.Tensor, percentiles, want_clipped: bool = False):
    """normalize_all on the GPU: CUDA float64 tensor -> uint8 alphabet indices (same shape)
    [and the clipped float64 tensor, the reference's first return value]."""
    if not (isinstance(signal, torch.Tensor) and signal.is_cuda and signal.dtype == torch.float64):
        raise TypeError("quantize needs a CUDA float64 tensor (no CPU fallback)")
    _lib.require_current(signal.device)
    x = signal.contiguous()
    sym = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    clipped = torch.empty_like(x) if want_clipped else None
    _lib.check(_lib.lib().ecgb_quantize_hip(_ptr(x), x.numel(), float(percentiles["percentile_1"]),
                                            float(percentiles["percentile_99"]), _ptr(sym), _ptr(clipped),
                                            _stream_ptr()))
    return (clipped, sym) if want_clipped else sym
