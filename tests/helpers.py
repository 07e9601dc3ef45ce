"""Shared test helpers: fixture loading and the CPU oracle handle."""
import json
import os
import pickle

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tokenizer(tag):
    with open(os.path.join(GOLDEN, f"tokenizer_{tag}.pkl"), "rb") as f:
        vocab, merges = pickle.load(f)
    with open(os.path.join(GOLDEN, f"percentiles_{tag}.json")) as f:
        pc = json.load(f)
    return vocab, merges, pc


def random_merges(rng, n_merges, alphabet=b"abcdef", max_len=12, dup_frac=0.05):
    """Random (possibly duplicate, possibly non-prefix-closed) merges list in reference shape."""
    merges = []
    for i in range(n_merges):
        if merges and rng.random() < dup_frac:
            seq = list(merges[rng.integers(len(merges))][0])
        else:
            ln = int(rng.integers(2, max_len + 1))
            seq = [int(alphabet[k]) for k in rng.integers(0, len(alphabet), size=ln)]
        merges.append((seq, 256 + i))
    return merges


def oracle_batch(trie, signal, pc):
    """Oracle ids for each record of a (B, ...) float64 numpy batch."""
    return [trie.quantize_encode(signal[b], pc["percentile_1"], pc["percentile_99"]) for b in range(signal.shape[0])]


class WordTokenizer:
    """A stand-in for the HF tokenizer surface the reference touches (main.py:141-150, data_loader.py:41-46,76-80,
    llm.py:33-36): whitespace words -> ids, `add_tokens`, special tokens, `batch_decode`."""

    def __init__(self, words=()):
        self.vocab = {}
        self.special = set()
        for w in ["<bos>", "<eos>"] + list(words):
            self.vocab.setdefault(w, len(self.vocab))
        self.special.update(["<bos>", "<eos>"])
        self.bos_token, self.eos_token, self.pad_token = "<bos>", "<eos>", None

    def __len__(self):
        return len(self.vocab)

    def add_tokens(self, toks, special_tokens=False):
        for t in toks:
            self.vocab.setdefault(t, len(self.vocab))
            if special_tokens:
                self.special.add(t)

    def add_special_tokens(self, d):
        self.pad_token = d["pad_token"]
        self.add_tokens([self.pad_token], special_tokens=True)

    @property
    def pad_token_id(self):
        return self.vocab[self.pad_token]

    @property
    def eos_token_id(self):
        return self.vocab[self.eos_token]

    def convert_tokens_to_ids(self, t):
        return [self.vocab[x] for x in t] if isinstance(t, (list, tuple)) else self.vocab[t]

    def __call__(self, texts, return_tensors="np", add_special_tokens=False):
        from types import SimpleNamespace
        for w in " ".join(texts).split():
            self.vocab.setdefault(w, len(self.vocab))
        return SimpleNamespace(input_ids=[np.array([self.vocab[w] for w in t.split()], dtype=np.int64) for t in texts])

    def batch_decode(self, ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        inv = {v: k for k, v in self.vocab.items()}
        out = []
        for row in ids.tolist() if hasattr(ids, "tolist") else ids:
            words = [inv[i] for i in row]
            out.append(" ".join(w for w in words if not (skip_special_tokens and w in self.special)))
        return out
