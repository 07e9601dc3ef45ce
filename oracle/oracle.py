"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  See oracle/ecgb_oracle.c for the reference file:line each function restates
and for the parity-pinning statement (trainer tie-breaks: parity unpinned).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libecgb_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ecgb_oracle.c", "ecgb_oracle_fasttrain.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libecgb_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, sz, u32p, u8p, dp = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        L.ecgb_oracle_quantize.argtypes = [dp, sz, C.c_double, C.c_double, dp, u8p]
        L.ecgb_oracle_quantize.restype = None
        L.ecgb_oracle_dequantize.argtypes = [u8p, sz, C.c_double, C.c_double, dp]
        L.ecgb_oracle_dequantize.restype = None
        L.ecgb_oracle_merge.argtypes = [u32p, sz, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ecgb_oracle_merge.restype = sz
        L.ecgb_oracle_get_stats.argtypes = [u32p, sz, u32p, sz]
        L.ecgb_oracle_get_stats.restype = sz
        for name in ("ecgb_oracle_bpe_train", "ecgb_oracle_bpe_train_fast"):
            f = getattr(L, name)
            f.argtypes = [u32p, C.POINTER(sz), C.c_uint32, u32p]
            f.restype = C.c_uint32
        L.ecgb_oracle_encode.argtypes = [u8p, sz, u32p, u32p, u32p, sz, u32p]
        L.ecgb_oracle_encode.restype = sz
        L.ecgb_oracle_trie_create.argtypes = [u32p, u32p, u32p, sz]
        L.ecgb_oracle_trie_create.restype = vp
        L.ecgb_oracle_trie_nodes.argtypes = [vp]
        L.ecgb_oracle_trie_nodes.restype = sz
        L.ecgb_oracle_trie_encode.argtypes = [vp, u8p, sz, u32p]
        L.ecgb_oracle_trie_encode.restype = sz
        L.ecgb_oracle_trie_destroy.argtypes = [vp]
        L.ecgb_oracle_trie_destroy.restype = None
        L.ecgb_oracle_quantize_encode.argtypes = [vp, dp, sz, C.c_double, C.c_double, u8p, u32p]
        L.ecgb_oracle_quantize_encode.restype = sz
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def flatten_merges(merges):
    """[(list[int] expansion, int id), ...] -> (flat u32, offsets u32[n+1], ids u32[n])."""
    n = len(merges)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    for i, (seq, _) in enumerate(merges):
        offsets[i + 1] = offsets[i] + len(seq)
    flat = np.empty(int(offsets[-1]), dtype=np.uint32)
    ids = np.empty(n, dtype=np.uint32)
    for i, (seq, tid) in enumerate(merges):
        flat[offsets[i]:offsets[i + 1]] = seq
        ids[i] = tid
    return flat, offsets, ids


def quantize(signal, p1, p99, want_clipped=False):
    """normalize_all restated: returns alphabet indices (uint8, same shape) [and clipped f64]."""
    x = np.ascontiguousarray(signal, dtype=np.float64)
    sym = np.empty(x.shape, dtype=np.uint8)
    clipped = np.empty(x.shape, dtype=np.float64) if want_clipped else None
    lib().ecgb_oracle_quantize(_p(x, C.c_double), x.size, float(p1), float(p99),
                               _p(clipped, C.c_double) if want_clipped else None,
                               _p(sym, C.c_uint8))
    return (clipped, sym) if want_clipped else sym


def quantize_python_style(signal, p1, p99) -> str:
    """normalize_all + the consumer's join, restated with the reference's COST structure (tokenizer_utils.py:14-19,
    data_loader.py:75): numpy arithmetic, then one Python-level call per sample (np.vectorize) to build a '<U1' array,
    then ''.join.  Only bench.py's cpu_baseline times it (the Python quantiser's cost is reported beside
    the encoder's); the C restatement `quantize` is the checker."""
    x = np.asarray(signal, dtype=np.float64)
    lo, hi = p1 - 0.5, p99 + 0.5
    norm = (x - lo) / ((hi - lo) + 1e-6)
    clipped = np.clip(norm, 0, 1)
    level = np.minimum(np.floor(clipped * 26), 25).astype(np.uint8)
    letters = "abcdefghijklmnopqrstuvwxyz"
    sym = np.vectorize(lambda v: letters[v])(level)
    return "".join(sym.flatten())


def dequantize(sym, p1, p99):
    s = np.ascontiguousarray(sym, dtype=np.uint8)
    out = np.empty(s.shape, dtype=np.float64)
    lib().ecgb_oracle_dequantize(_p(s, C.c_uint8), s.size, float(p1), float(p99), _p(out, C.c_double))
    return out


def symbols_to_text(sym) -> bytes:
    return (np.asarray(sym, dtype=np.uint8).reshape(-1) + ord("a")).astype(np.uint8).tobytes()


def merge(ids, left, right, new_id):
    a = np.array(ids, dtype=np.uint32)
    n = lib().ecgb_oracle_merge(_p(a, C.c_uint32), a.size, left, right, new_id)
    return a[:n].tolist()


def get_stats(ids):
    a = np.array(ids, dtype=np.uint32)
    cap = max(1, a.size)
    out = np.zeros(3 * cap, dtype=np.uint32)
    n = lib().ecgb_oracle_get_stats(_p(a, C.c_uint32), a.size, _p(out, C.c_uint32), cap)
    return {(int(out[3 * i]), int(out[3 * i + 1])): int(out[3 * i + 2]) for i in range(n)}


def byte_to_string(b: int) -> str:
    """lib.rs:50-56."""
    return chr(b) if b <= 127 else f"<{b}>"


def pairs_to_vocab_merges(pairs):
    """lib.rs:73-75,101-110: vocab strings and byte expansions from the chosen pairs."""
    vocab = {i: byte_to_string(i) for i in range(256)}
    expand = {i: [i] for i in range(256)}
    merges = []
    for i, (l, r) in enumerate(pairs):
        nid = 256 + i
        vocab[nid] = vocab[l] + vocab[r]
        expand[nid] = expand[l] + expand[r]
        merges.append((list(expand[nid]), nid))
    return vocab, merges


def byte_pair_encoding(text, num_merges, fast=True):
    """rust_bpe.byte_pair_encoding restated -> (ids, vocab, merges).  `text`: str or bytes."""
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    ids = np.frombuffer(raw, dtype=np.uint8).astype(np.uint32)
    n = C.c_size_t(ids.size)
    pairs = np.zeros(2 * max(1, num_merges), dtype=np.uint32)
    fn = lib().ecgb_oracle_bpe_train_fast if fast else lib().ecgb_oracle_bpe_train
    done = fn(_p(ids, C.c_uint32), C.byref(n), num_merges, _p(pairs, C.c_uint32))
    if done == 0xFFFFFFFF:
        raise MemoryError("oracle trainer allocation failure")
    pl = [(int(pairs[2 * i]), int(pairs[2 * i + 1])) for i in range(done)]
    vocab, merges = pairs_to_vocab_merges(pl)
    return ids[: n.value].tolist(), vocab, merges


def encode_text(text, merges):
    """rust_bpe.encode_text restated (trie rebuilt per call, as the reference does)."""
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    t = np.frombuffer(raw, dtype=np.uint8)
    flat, offsets, ids = flatten_merges(merges)
    out = np.empty(max(1, t.size), dtype=np.uint32)
    n = lib().ecgb_oracle_encode(_p(t, C.c_uint8), t.size, _p(flat, C.c_uint32),
                                 _p(offsets, C.c_uint32), _p(ids, C.c_uint32), len(merges),
                                 _p(out, C.c_uint32))
    if n == C.c_size_t(-1).value:
        raise MemoryError("oracle encode allocation failure")
    return out[:n].tolist()


def decode_text(encoded_ids, vocab):
    """tokenizer_utils.py:75-77."""
    return "".join(vocab[i] for i in encoded_ids)


class Trie:
    """Build-once trie handle (CPU-baseline variant)."""

    def __init__(self, merges):
        self._flat, self._off, self._ids = flatten_merges(merges)
        self._h = lib().ecgb_oracle_trie_create(_p(self._flat, C.c_uint32), _p(self._off, C.c_uint32),
                                                _p(self._ids, C.c_uint32), len(merges))
        if not self._h:
            raise MemoryError("oracle trie allocation failure")

    @property
    def n_nodes(self):
        return lib().ecgb_oracle_trie_nodes(self._h)

    def encode_bytes(self, raw) -> np.ndarray:
        t = np.frombuffer(bytes(raw), dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
        out = np.empty(max(1, t.size), dtype=np.uint32)
        n = lib().ecgb_oracle_trie_encode(self._h, _p(t, C.c_uint8), t.size, _p(out, C.c_uint32))
        return out[:n]

    def quantize_encode(self, signal, p1, p99) -> np.ndarray:
        """data_loader.py:74-76 for one (12,L) float64 record."""
        x = np.ascontiguousarray(signal, dtype=np.float64)
        scratch = np.empty(x.size, dtype=np.uint8)
        out = np.empty(max(1, x.size), dtype=np.uint32)
        n = lib().ecgb_oracle_quantize_encode(self._h, _p(x, C.c_double), x.size, float(p1), float(p99),
                                              _p(scratch, C.c_uint8), _p(out, C.c_uint32))
        return out[:n]

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.ecgb_oracle_trie_destroy(self._h)
            except Exception:
                pass
            self._h = None
