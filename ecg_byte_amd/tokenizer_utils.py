"""Mirror of the reference's ecg_byte/utils/tokenizer_utils.py (same names, argument meaning and
return shapes) with the compute on the MI355X.  Analysis/plot helpers of the reference
(analyze_token_distribution, track_encoding) are out of scope.
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from . import rust_bpe
from .tokenizer import quantize as _quantize_dev

ALPHABET = list("abcdefghijklmnopqrstuvwxyz")          # tokenizer_utils.py:12
_ALPHABET_ARR = np.array(ALPHABET, dtype="<U1")


def normalize_all(signal, percentiles):
    """tokenizer_utils.py:14-19 -> (clipped_normalized float64, symbol_signal '<U1'), same shape.

    Accepts a numpy array (any float dtype; promoted to float64 as NumPy 2 does with the
    float64 percentile scalars) or a CUDA float64 tensor."""
    if isinstance(signal, torch.Tensor):
        x = signal.to(device="cuda", dtype=torch.float64)
    else:
        x = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float64)).cuda()
    clipped, sym = _quantize_dev(x, percentiles, want_clipped=True)
    return clipped.cpu().numpy(), _ALPHABET_ARR[sym.cpu().numpy()]


def quantize_symbols(signal, percentiles) -> np.ndarray:
    """Alphabet indices (uint8) only -- what every consumer of normalize_all actually uses."""
    x = signal if isinstance(signal, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(signal, dtype=np.float64))
    return _quantize_dev(x.to(device="cuda", dtype=torch.float64), percentiles).cpu().numpy()


def reverse_normalize_all(symbol_signal, percentiles):
    """tokenizer_utils.py:22-28 (not an exact inverse: index/25 * (max-min) + min)."""
    min_vals = percentiles["percentile_1"] - 0.5
    max_vals = percentiles["percentile_99"] + 0.5
    s = np.asarray(symbol_signal)
    idx = (s.view(np.uint32).reshape(s.shape) - ord("a")) if s.dtype.kind == "U" else s
    clipped_normalized = idx.astype(np.float64) / (len(ALPHABET) - 1)
    return clipped_normalized * (max_vals - min_vals) + min_vals


def encode_text(text, merges):
    """tokenizer_utils.py:71-73."""
    return rust_bpe.encode_text(text, merges)


def decode_text(encoded_ids, vocab):
    """tokenizer_utils.py:75-77."""
    return "".join(vocab[i] for i in encoded_ids)


def save_vocab_and_merges(vocab, merges, filename):
    """tokenizer_utils.py:62-64: pickle of the 2-tuple."""
    with open(filename, "wb") as f:
        pickle.dump((vocab, merges), f)


def load_vocab_and_merges(filename):
    """tokenizer_utils.py:66-69."""
    with open(filename, "rb") as f:
        vocab, merges = pickle.load(f)
    return vocab, merges


def process_ecg(ecg, percentiles):
    """tokenizer_utils.py:56-59: path to a (12, L) .npy -> lead-major symbol string."""
    sig = np.load(ecg)
    return (quantize_symbols(sig, percentiles).reshape(-1) + ord("a")).astype(np.uint8).tobytes().decode("ascii")


def process_large_file(file_path, percentiles, num_processes=1, n=None):
    """tokenizer_utils.py:79-93: concatenated symbol strings of the listed .npy files, in file
    order.  `num_processes` is accepted for compatibility; records are quantised on the GPU in
    batches instead of in a process pool."""
    paths = []
    with open(file_path, "r") as f:
        for i, line in enumerate(f):
            if n is not None and i >= n:
                break
            paths.append(line.strip())
    return process_paths(paths, percentiles).decode("ascii")


def process_paths(paths, percentiles) -> bytes:
    """Concatenated symbol bytes ('a'..'z') of the given .npy records, in order; records are quantised on the GPU in batches."""
    parts = []
    batch, shape = [], None
    def flush():
        if batch:
            x = torch.from_numpy(np.stack(batch)).cuda()
            s = _quantize_dev(x, percentiles).cpu().numpy().reshape(-1)
            parts.append((s + ord("a")).astype(np.uint8).tobytes())
            batch.clear()
    for p in paths:
        sig = np.ascontiguousarray(np.load(p), dtype=np.float64)
        if shape is not None and sig.shape != shape:
            flush()
        shape = sig.shape
        batch.append(sig)
        if len(batch) >= 256:
            flush()
    flush()
    return b"".join(parts)
