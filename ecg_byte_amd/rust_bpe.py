"""Drop-in for the reference's PyO3 module `rust_bpe` (ecg_byte/rust_bpe/src/lib.rs:195-200):

    byte_pair_encoding(text, num_merges, num_threads) -> (ids, vocab, merges)
    encode_text(text, merges) -> list[int]

Same positional signatures, same return shapes; the work runs on the MI355X through the C
ABI.  `sys.modules['rust_bpe'] = ecg_byte_amd.rust_bpe` makes the reference's
tokenizer_utils.py / train_tokenizer.py use it unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from .tokenizer import HipTokenizer

_CACHE: dict = {}
_CACHE_MAX = 8


def _content_key(merges) -> bytes:
    """Digest of the WHOLE list (every expansion element and id).  The reference rebuilds its trie from `merges` on every call
    (lib.rs:153-161), so any mutation -- a replaced entry, an inner list edited in place -- is honoured there; a cache keyed on the
    object's identity or on a few sampled fields would keep serving the stale trie.  Serialising 4 000 entries and hashing them takes
    ~1.5 ms, less than the reference's own per-call rebuild.  marshal format 2: later formats write back-references that depend on object
    identity and reference counts, so equal lists could serialise differently (spurious misses); format 2 is a function of the content."""
    import hashlib
    import marshal
    try:
        blob = marshal.dumps(merges, 2)
    except ValueError:           # numpy integers and other non-marshallable element types
        import pickle
        blob = pickle.dumps([(list(map(int, seq)), int(tid)) for seq, tid in merges], protocol=5)
    return hashlib.blake2b(blob, digest_size=16).digest()


def tokenizer_for(merges) -> HipTokenizer:
    """The reference rebuilds the trie on every encode_text call (lib.rs:153-161); here the device handle is cached per CONTENT of
    the merges list."""
    key = _content_key(merges)
    tk = _CACHE.get(key)
    if tk is not None:
        return tk
    tk = HipTokenizer(merges)
    if len(_CACHE) >= _CACHE_MAX:
        _CACHE.pop(next(iter(_CACHE)))
    _CACHE[key] = tk
    return tk


def encode_text(text, merges):
    """rust_bpe.encode_text (lib.rs:149-193): greedy longest-match token ids of `text`."""
    if isinstance(text, str):
        raw = text.encode("utf-8")          # text.as_bytes(), lib.rs:151
    elif isinstance(text, (bytes, bytearray, memoryview)):
        raw = bytes(text)
    else:
        raise TypeError("argument 'text': expected str")   # PyO3 raises TypeError too
    tk = tokenizer_for(merges)
    if len(raw) == 0:
        return []
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    ids, counts = tk.encode_bytes(t[None])
    n = int(counts[0].item())
    return ids[0, :n].cpu().tolist()


def byte_to_string(b: int) -> str:
    """lib.rs:50-56."""
    return chr(b) if b <= 127 else f"<{b}>"


def vocab_merges_from_pairs(pairs):
    """vocab / merges bookkeeping of lib.rs:73-75,101-110 from the chosen (left,right) pairs."""
    vocab = {i: byte_to_string(i) for i in range(256)}
    expand = {i: [i] for i in range(256)}
    merges = []
    for i, (l, r) in enumerate(pairs):
        nid = 256 + i
        l, r = int(l), int(r)
        vocab[nid] = vocab[l] + vocab[r]
        expand[nid] = expand[l] + expand[r]
        merges.append((list(expand[nid]), nid))
    return vocab, merges


def byte_pair_encoding(text, num_merges, num_threads=1):
    """rust_bpe.byte_pair_encoding (lib.rs:58-125) on the GPU.  `num_threads` is accepted for
    signature compatibility (the reference sizes its rayon pool with it) and ignored.
    Tie-break among equal-count pairs: numerically smallest (left, right) -- the reference's is
    hash/schedule dependent."""
    from .trainer import bpe_train
    if isinstance(text, str):
        raw = text.encode("utf-8")
    else:
        raw = bytes(text)
    ids, pairs = bpe_train(np.frombuffer(raw, dtype=np.uint8), int(num_merges))
    vocab, merges = vocab_merges_from_pairs(pairs)
    return ids, vocab, merges
