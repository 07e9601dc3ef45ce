"""A SECOND, independent restatement of the reference's tokenizer in pure Python -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

oracle/ecgb_oracle.c restates ecg_byte/rust_bpe/src/lib.rs with its own data structures (an open-addressing edge map, an
incremental trainer).  This file transliterates the same Rust statement by statement, with dict-of-dict tries and plain lists,
so that a slip in either restatement shows up as a disagreement between the two (tests/test_oracle.py fuzzes one against the
other).  It does NOT lift the "parity unpinned" status of the encoder: neither restatement has been checked against the compiled
Rust crate (no cargo / rustc in the build container, and the reference ships no test vectors for this path).

    merge            lib.rs:10-26
    get_stats        lib.rs:28-48   (the sequential branch; the rayon branch computes the same multiset)
    byte_to_string   lib.rs:50-56
    TrieNode         lib.rs:127-147
    encode_text      lib.rs:149-193
    byte_pair_encoding  lib.rs:58-125 with the tie-break this repository DEFINES (numerically smallest (left, right) among the
                     pairs of maximal count; the reference's own choice depends on hash-map iteration order and thread schedule)
"""


def merge(ids, pair, new_id):
    """lib.rs:10-26 -- in-place left-to-right compaction, returned as a new list."""
    out = []
    i = 0
    while i < len(ids):
        if i + 1 < len(ids) and (ids[i], ids[i + 1]) == pair:
            out.append(new_id)
            i += 2
        else:
            out.append(ids[i])
            i += 1
    return out


def get_stats(ids):
    """lib.rs:28-48 -- every window of two counts, overlapping ones included ("aaa" has (a, a) twice)."""
    acc = {}
    for k in range(len(ids) - 1):
        w = (ids[k], ids[k + 1])
        acc[w] = acc.get(w, 0) + 1
    return acc


def byte_to_string(b):
    """lib.rs:50-56"""
    return chr(b) if b <= 127 else f"<{b}>"


class TrieNode:
    """lib.rs:127-147"""

    def __init__(self):
        self.children = {}
        self.token_id = None

    def insert(self, token, token_id):
        node = self
        for i in token:
            nxt = node.children.get(i)
            if nxt is None:
                nxt = node.children[i] = TrieNode()
            node = nxt
        node.token_id = token_id          # a later insert of the same sequence overwrites (lib.rs:145)


def encode_text(text, merges):
    """lib.rs:149-193.  text: bytes or str."""
    ids = list(text.encode("utf-8") if isinstance(text, str) else bytes(text))
    root = TrieNode()
    for b in range(256):
        root.insert([b], b)
    for token_sequence, token_id in merges:
        root.insert(token_sequence, token_id)
    output_ids = []
    i = 0
    while i < len(ids):
        node = root
        match_len = 0
        match_id = None
        for j in range(i, len(ids)):
            child = node.children.get(ids[j])
            if child is None:
                break
            node = child
            if node.token_id is not None:
                match_len = j - i + 1
                match_id = node.token_id
        if match_id is not None:
            output_ids.append(match_id)
            i += match_len
        else:
            output_ids.append(ids[i])
            i += 1
    return output_ids


def byte_pair_encoding(text, num_merges):
    """lib.rs:58-125 with the defined tie-break.  Returns (ids, vocab, merges) in the reference's shapes."""
    ids = list(text.encode("utf-8") if isinstance(text, str) else bytes(text))
    vocab = {i: byte_to_string(i) for i in range(256)}
    vocab_bytes = {i: [i] for i in range(256)}
    merges = []
    for i in range(num_merges):
        pairs = get_stats(ids)
        if not pairs:
            break                                                    # lib.rs:88-90
        top = max(pairs.values())
        best = min(p for p, c in pairs.items() if c == top)          # DEFINED tie-break
        new_id = 256 + i                                             # lib.rs:97
        ids = merge(ids, best, new_id)
        vocab[new_id] = vocab[best[0]] + vocab[best[1]]
        vocab_bytes[new_id] = vocab_bytes[best[0]] + vocab_bytes[best[1]]
        merges.append((list(vocab_bytes[new_id]), new_id))
    return ids, vocab, merges
