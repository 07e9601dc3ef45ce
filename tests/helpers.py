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
