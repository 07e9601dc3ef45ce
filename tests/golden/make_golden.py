#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.  Runs ONLY in the build container
(it imports the Python reference from /root/reference, which never travels to the GPU box).

Fixtures written (data only -- inputs and expected outputs):
  quantize_ref.npz      inputs + outputs of the REFERENCE normalize_all / reverse_normalize_all
                        (ecg_byte/utils/tokenizer_utils.py:14-28), incl. values at bin edges
  assemble_ref.json     inputs + outputs of the REFERENCE ECGTokenDataset._prepare_training /
                        _prepare_inference (ecg_byte/data_loader.py:91-132) with a stand-in
                        HF tokenizer object (special ids only)
  percentiles_c{1,2}.json  global percentile dicts of the synthetic generator
  tokenizer_c{1,2}.pkl  (vocab, merges) pickles in the reference's on-disk format
                        (tokenizer_utils.py:62-69), trained by the ORACLE trainer on 2000
                        synthetic ECGs, seed 1: C1 = 12x1000 / 1000 merges, C2 = 12x5000 / 4000
  encode_oracle.npz     ORACLE token ids of a few synthetic records (regression anchor; the
                        Rust reference cannot be built here, see oracle/ecgb_oracle.c header)

Usage:  python tests/golden/make_golden.py [--skip-tokenizers]
"""
import argparse
import json
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ecg_byte_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def import_reference():
    sys.path.insert(0, "/root/reference")
    sys.modules.setdefault("rust_bpe", types.ModuleType("rust_bpe"))  # module-level import in tokenizer_utils.py:2
    from ecg_byte.utils import tokenizer_utils as ref_tu
    from ecg_byte import data_loader as ref_dl
    return ref_tu, ref_dl


def make_quantize(ref_tu):
    rng = np.random.default_rng(1234)
    cases = []
    params = [(-1.0, 1.0), (-0.07743950602542989, 0.8725949522194346), (-3.25, 7.5), (0.0, 0.0),
              (1e6, 1e6 + 2.0), (-1e-3, 1e-3)]
    for p1, p99 in params:
        a = p1 - 0.5
        d = ((p99 + 0.5) - (p1 - 0.5)) + 1e-6
        xs = [rng.normal((p1 + p99) / 2, (p99 - p1 + 1.0) / 2, size=2000),
              rng.uniform(a - 1.0, a + d + 1.0, size=2000)]
        # values straddling every bin edge: a + d*k/26 and +-1..3 ulp around it
        edges = a + d * np.arange(0, 27) / 26.0
        near = []
        for e in edges:
            v = e
            for _ in range(4):
                v = np.nextafter(v, -np.inf)
            for _ in range(9):
                near.append(v)
                v = np.nextafter(v, np.inf)
        xs.append(np.array(near))
        xs.append(np.array([-np.inf, np.inf, -1e300, 1e300, 0.0, -0.0, a, a + d, p1, p99]))
        x = np.concatenate(xs).astype(np.float64)
        x = x[: (x.size // 12) * 12].reshape(12, -1)
        clipped, symbols = ref_tu.normalize_all(x, {"percentile_1": p1, "percentile_99": p99})
        idx = (symbols.view(np.uint32).reshape(symbols.shape) - ord("a")).astype(np.uint8)
        back = ref_tu.reverse_normalize_all(symbols, {"percentile_1": p1, "percentile_99": p99})
        cases.append((p1, p99, x, clipped, idx, back))
    out = {"n_cases": np.array(len(cases))}
    for i, (p1, p99, x, clipped, idx, back) in enumerate(cases):
        out[f"p_{i}"] = np.array([p1, p99])
        out[f"x_{i}"] = x
        out[f"clipped_{i}"] = clipped
        out[f"sym_{i}"] = idx
        out[f"back_{i}"] = back
    np.savez_compressed(os.path.join(HERE, "quantize_ref.npz"), **out)
    print("quantize_ref.npz:", len(cases), "cases")


class _FakeHF:
    """Stand-in for the HF tokenizer surface ECGTokenDataset.__init__ touches."""
    pad_token, bos_token, eos_token = "<pad>", "<bos>", "<eos>"

    def __init__(self, ids):
        self._ids = ids

    def convert_tokens_to_ids(self, t):
        if isinstance(t, list):
            return [self._ids[x] for x in t]
        return self._ids[t]


def make_assemble(ref_dl):
    rng = np.random.default_rng(7)
    special = {"<pad>": 128259, "<bos>": 128000, "<eos>": 128001, "<sig_start>": 132012, "<sig_end>": 132013}
    cases = []
    for pad_to_max, n_sig, n_q, n_a in [(1020, 4520, 12, 9), (1020, 300, 20, 31), (1020, 990, 10, 20),
                                        (64, 40, 8, 16), (64, 39, 8, 16), (64, 41, 8, 16), (64, 0, 8, 16),
                                        (32, 100, 16, 16), (16, 5, 3, 2), (2044, 1500, 24, 4),
                                        (48, 10, 1, 1), (48, 200, 0, 0)]:
        ds = ref_dl.ECGTokenDataset.__new__(ref_dl.ECGTokenDataset)
        ds.args = types.SimpleNamespace(pad_to_max=pad_to_max, inference=False)
        ds.pad_id, ds.bos_id, ds.eos_id = special["<pad>"], special["<bos>"], special["<eos>"]
        ds.sig_start_id, ds.sig_end_id = [special["<sig_start>"]], [special["<sig_end>"]]
        sig = rng.integers(128260, 132012, size=n_sig).tolist()
        q = rng.integers(1000, 100000, size=n_q).tolist()
        a = rng.integers(1000, 100000, size=n_a).tolist()
        r = ds._prepare_training(list(sig), list(q), list(a), None, None, None)
        inf = ds._prepare_inference(list(sig), list(q), "answer", "question")
        cases.append({
            "pad_to_max": pad_to_max, "special": special, "sig": sig, "q": q, "a": a,
            "tokenized_signal": r["tokenized_signal"].tolist(),
            "attn_mask": r["attn_mask"].tolist(),
            "quantized_signal_ids_input": r["quantized_signal_ids_input"].tolist(),
            "position_ids": r["position_ids"].tolist(),
            "inference_tokenized_signal": inf["tokenized_signal"].tolist(),
            "inference_attn_mask": inf["attn_mask"].tolist(),
        })
    with open(os.path.join(HERE, "assemble_ref.json"), "w") as f:
        json.dump(cases, f)
    print("assemble_ref.json:", len(cases), "cases")


def make_tokenizer(tag, L, num_merges, n_train=2000):
    pc = synth.synth_percentiles(L, seed=0)
    with open(os.path.join(HERE, f"percentiles_{tag}.json"), "w") as f:
        json.dump(pc, f)
    x = synth.synth_ecg(n_train, L, seed=1)
    sym = O.quantize(x, pc["percentile_1"], pc["percentile_99"])
    del x
    text = O.symbols_to_text(sym)
    del sym
    ids, vocab, merges = O.byte_pair_encoding(text, num_merges, fast=True)
    with open(os.path.join(HERE, f"tokenizer_{tag}.pkl"), "wb") as f:
        pickle.dump((vocab, merges), f, protocol=4)
    print(f"tokenizer_{tag}.pkl: {len(merges)} merges, corpus {len(text)} -> {len(ids)} ids "
          f"({len(text) / len(ids):.2f}x)")
    return pc, vocab, merges


def make_encode_oracle(toks):
    out = {}
    for tag, (L, pc, merges) in toks.items():
        x = synth.synth_ecg(3, L, seed=0)
        trie = O.Trie(merges)
        for b in range(3):
            out[f"{tag}_ids_{b}"] = trie.quantize_encode(x[b], pc["percentile_1"], pc["percentile_99"]).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "encode_oracle.npz"), **out)
    print("encode_oracle.npz:", {k: v.size for k, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-tokenizers", action="store_true")
    args = ap.parse_args()
    O.build()
    ref_tu, ref_dl = import_reference()
    make_quantize(ref_tu)
    make_assemble(ref_dl)
    if not args.skip_tokenizers:
        toks = {}
        pc, _, merges = make_tokenizer("c1", 1000, 1000)
        toks["c1"] = (1000, pc, merges)
        pc, _, merges = make_tokenizer("c2", 5000, 4000)
        toks["c2"] = (5000, pc, merges)
        make_encode_oracle(toks)


if __name__ == "__main__":
    main()
