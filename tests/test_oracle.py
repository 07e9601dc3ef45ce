"""CPU tests of the oracle itself: it is pinned here against (i) outputs of the Python
reference captured in tests/golden (quantiser, sequence assembly), (ii) the hand-derived
known-answer vectors of SURVEY.md §8c (the Rust reference ships no tests and cannot be built
in this environment), (iii) the reference's own round-trip property
(ecg_byte/train_tokenizer.py:58-60)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_tokenizer, random_merges
from oracle import assemble as OA
from oracle import oracle as O


# ---- encode_text known-answer vectors (lib.rs:149-193 semantics) ---------------------------
def test_encode_longest_match_not_merge_order():
    # merge-order BPE would give [97, 256]; the trie's longest match gives [257, 99]
    assert O.encode_text("abc", [([98, 99], 256), ([97, 98], 257)]) == [257, 99]


def test_encode_runs():
    m = [([97, 97], 256)]
    assert O.encode_text("aaaa", m) == [256, 256]
    assert O.encode_text("aaa", m) == [256, 97]


def test_encode_duplicate_expansion_last_wins():
    assert O.encode_text("abc", [([97, 98, 99], 256), ([97, 98, 99], 257)]) == [257]


def test_encode_interior_nodes_need_not_be_tokens():
    m = [([97, 98, 99, 100], 256)]
    assert O.encode_text("abcd", m) == [256]
    assert O.encode_text("abcx", m) == [97, 98, 99, 120]


def test_encode_empty_and_no_merges():
    assert O.encode_text("", [([97, 97], 256)]) == []
    assert O.encode_text("hello", []) == list(b"hello")
    assert O.encode_text("hé", []) == list("hé".encode("utf-8"))   # text.as_bytes()


# ---- merge / get_stats / trainer (lib.rs:10-48, 58-125) -------------------------------------
def test_merge_and_get_stats():
    assert O.merge([97, 97, 97], 97, 97, 256) == [256, 97]
    assert O.merge([1, 2, 1, 2, 3], 1, 2, 9) == [9, 9, 3]
    assert O.merge([], 1, 2, 9) == []
    assert O.get_stats([97, 97, 97]) == {(97, 97): 2}     # overlapping windows both count
    assert O.get_stats([97]) == {}


def test_trainer_known_answer():
    ids, vocab, merges = O.byte_pair_encoding("aaabdaaabac", 1, fast=False)
    assert ids == [256, 97, 98, 100, 256, 97, 98, 97, 99]        # step 0 picks (97,97), count 4
    assert merges == [([97, 97], 256)] and vocab[256] == "aa"
    # step 1 is a genuine tie ((97,98) vs (256,97), both 2): the DEFINED tie-break is the smallest pair
    ids2, _, merges2 = O.byte_pair_encoding("aaabdaaabac", 2, fast=False)
    assert merges2[1] == ([97, 98], 257)
    assert ids2 == [256, 257, 100, 256, 257, 97, 99]


def test_trainer_stops_when_no_pairs():
    ids, vocab, merges = O.byte_pair_encoding("ab", 5, fast=False)
    assert ids == [256] and len(merges) == 1            # lib.rs:88-90
    assert O.byte_pair_encoding("", 3, fast=True)[2] == []
    assert O.byte_pair_encoding("a", 3, fast=True)[0] == [97]


@pytest.mark.parametrize("seed", range(6))
def test_fast_trainer_equals_literal_loop(seed):
    rng = np.random.default_rng(seed)
    for _ in range(40):
        n = int(rng.integers(0, 600))
        k = int(rng.choice([1, 2, 3, 5, 26]))
        text = bytes(rng.integers(97, 97 + k, size=n).astype(np.uint8))
        nm = int(rng.integers(0, 80))
        assert O.byte_pair_encoding(text, nm, fast=True) == O.byte_pair_encoding(text, nm, fast=False)


def test_trainer_is_valid_greedy():
    """Properties the reference guarantees whatever its tie-break: every chosen pair has the
    maximal count at its step, ids are 256+i, merge() semantics, decode(ids) == text."""
    rng = np.random.default_rng(11)
    text = bytes(rng.choice(np.frombuffer(b"aaaabbbcd", dtype=np.uint8), size=3000))
    ids, vocab, merges = O.byte_pair_encoding(text, 50, fast=True)
    cur = list(text)
    for i, (exp, nid) in enumerate(merges):
        assert nid == 256 + i
        stats = O.get_stats(cur)
        best = max(stats.values())
        # recover (left,right) of this merge from the expansions
        cands = [(l, r) for (l, r), c in stats.items() if c == best]
        chosen = min(cands)
        cur = O.merge(cur, chosen[0], chosen[1], nid)
        expand = {**{b: [b] for b in range(256)}, **{m[1]: m[0] for m in merges[:i]}}
        assert expand[chosen[0]] + expand[chosen[1]] == exp
    assert cur == ids
    assert O.decode_text(ids, vocab) == text.decode("ascii")


# ---- quantiser vs the Python reference (tokenizer_utils.py:14-28) ---------------------------
def test_quantiser_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "quantize_ref.npz"))
    for i in range(int(z["n_cases"])):
        p1, p99 = z[f"p_{i}"]
        clipped, sym = O.quantize(z[f"x_{i}"], p1, p99, want_clipped=True)
        assert np.array_equal(sym, z[f"sym_{i}"]), f"case {i}: symbols differ"
        assert np.array_equal(clipped, z[f"clipped_{i}"]), f"case {i}: clipped differs"   # bit-exact
        assert np.array_equal(O.dequantize(z[f"sym_{i}"], p1, p99), z[f"back_{i}"])


def test_quantiser_hand_anchors():
    sym = O.quantize(np.array([0.0, 1.5, -2.0, 2.0]), -1.0, 1.0)
    assert "".join(chr(97 + s) for s in sym) == "mzaz"


# ---- sequence assembly vs the Python reference (data_loader.py:91-132) ----------------------
def test_assemble_matches_reference_golden():
    with open(os.path.join(GOLDEN, "assemble_ref.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 12
    for c in cases:
        sp = c["special"]
        r = OA.prepare_training(c["sig"], c["q"], c["a"], sp["<pad>"], sp["<bos>"], sp["<eos>"],
                                sp["<sig_start>"], sp["<sig_end>"], c["pad_to_max"])
        for k in ("tokenized_signal", "attn_mask", "quantized_signal_ids_input", "position_ids"):
            assert np.array_equal(r[k], np.asarray(c[k])), (c["pad_to_max"], k)
        inf = OA.prepare_inference(c["sig"], c["q"], sp["<pad>"], sp["<bos>"], sp["<sig_start>"], sp["<sig_end>"])
        assert np.array_equal(inf["tokenized_signal"], np.asarray(c["inference_tokenized_signal"]))
        assert np.array_equal(inf["attn_mask"], np.asarray(c["inference_attn_mask"]))


# ---- fixture tokenizers: round trip + regression anchor -------------------------------------
@pytest.mark.parametrize("tag,L", [("c1", 1000), ("c2", 5000)])
def test_fixture_tokenizer_roundtrip(tag, L):
    from ecg_byte_amd import synth
    vocab, merges, pc = load_tokenizer(tag)
    assert len(merges) == {"c1": 1000, "c2": 4000}[tag]
    assert all(m[1] == 256 + i for i, m in enumerate(merges))
    x = synth.synth_ecg(3, L, seed=0)
    z = np.load(os.path.join(GOLDEN, "encode_oracle.npz"))
    trie = O.Trie(merges)
    for b in range(3):
        text = O.symbols_to_text(O.quantize(x[b], pc["percentile_1"], pc["percentile_99"]))
        ids = O.encode_text(text, merges)                       # trie rebuilt per call
        assert O.decode_text(ids, vocab) == text.decode("ascii")  # train_tokenizer.py:58-60
        assert np.array_equal(trie.encode_bytes(text), ids)     # build-once handle agrees
        assert np.array_equal(z[f"{tag}_ids_{b}"], ids)         # committed regression anchor


def test_random_merges_roundtrip():
    rng = np.random.default_rng(5)
    for _ in range(20):
        merges = random_merges(rng, int(rng.integers(1, 200)))
        vocab = {i: O.byte_to_string(i) for i in range(256)}
        for seq, tid in merges:
            vocab[tid] = "".join(chr(b) for b in seq)
        text = bytes(rng.integers(97, 103, size=int(rng.integers(0, 2000))).astype(np.uint8))
        assert O.decode_text(O.encode_text(text, merges), vocab) == text.decode("ascii")


# ---- the two restatements of lib.rs against each other ---------------------------------------------------------------------
def test_second_restatement_agrees_on_the_known_answer_vectors():
    from oracle import lib_rs_literal as R
    assert R.encode_text("abc", [([98, 99], 256), ([97, 98], 257)]) == [257, 99]
    assert R.encode_text("aaaa", [([97, 97], 256)]) == [256, 256] and R.encode_text("aaa", [([97, 97], 256)]) == [256, 97]
    assert R.encode_text("abc", [([97, 98, 99], 256), ([97, 98, 99], 257)]) == [257]
    assert R.encode_text("abcd", [([97, 98, 99, 100], 256)]) == [256] and R.encode_text("abcx", [([97, 98, 99, 100], 256)]) == [97, 98, 99, 120]
    assert R.merge([97, 97, 97], (97, 97), 256) == [256, 97] and R.get_stats([97, 97, 97]) == {(97, 97): 2}
    ids, vocab, merges = R.byte_pair_encoding("aaabdaaabac", 1)
    assert ids == [256, 97, 98, 100, 256, 97, 98, 97, 99] and merges == [([97, 97], 256)] and vocab[256] == "aa"


@pytest.mark.parametrize("seed", range(8))
def test_differential_fuzz_c_oracle_vs_literal_python_restatement(seed):
    """oracle/ecgb_oracle.c (the parity anchor of the HIP encoder and trainer) against oracle/lib_rs_literal.py, a statement-by-statement
    Python transliteration of lib.rs:10-56,127-193 written independently of it: random vocabularies (duplicates, non-prefix-closed
    sets, bytes outside every merge), random texts, trained tokenizers.  Encode ids, trainer ids / vocab / merges must be identical."""
    from oracle import lib_rs_literal as R
    rng = np.random.default_rng(1000 + seed)
    for _ in range(25):
        alphabet = bytes(rng.choice(np.arange(97, 123), size=int(rng.integers(1, 7)), replace=False).astype(np.uint8))
        merges = random_merges(rng, int(rng.integers(0, 60)), alphabet=alphabet, max_len=int(rng.integers(2, 14)), dup_frac=0.1)
        extra = alphabet + (b"" if rng.random() < 0.5 else bytes([200, 10]))
        text = bytes(rng.choice(np.frombuffer(extra, dtype=np.uint8), size=int(rng.integers(0, 600))))
        assert list(O.encode_text(text, merges)) == R.encode_text(text, merges)
    for _ in range(6):
        k = int(rng.choice([1, 2, 3, 5]))
        text = bytes(rng.integers(97, 97 + k, size=int(rng.integers(0, 400))).astype(np.uint8))
        nm = int(rng.integers(0, 40))
        ids, vocab, merges = R.byte_pair_encoding(text, nm)
        for fast in (False, True):
            got = O.byte_pair_encoding(text, nm, fast=fast)
            assert got[0] == ids and got[1] == vocab and got[2] == merges, (seed, k, nm, fast)
        assert list(O.encode_text(text, merges)) == R.encode_text(text, merges)
        assert "".join(vocab[i] for i in R.encode_text(text, merges)) == text.decode()
