"""GPU parity tests (run with -m gpu on the MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden fixtures, and
-- at BASELINE's full size -- through size-independent properties.  Bar: bit-exact."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_tokenizer, oracle_batch, random_merges
from oracle import assemble as OA
from oracle import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ecg_byte_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "HIP extension not built -- no fallback exists"
    return torch.device("cuda", 0)


@pytest.fixture(params=[0, 1, 2, 3, 4], ids=["plan-auto", "plan-workgroup", "plan-wave", "plan-wave-long-segments", "plan-lane-per-chunk"])
def plan(request, dev):
    """Runs a test under each encode kernel: automatic choice, workgroup-per-stream, wave-per-stream."""
    from ecg_byte_amd.tokenizer import set_encode_plan
    set_encode_plan(request.param)
    yield request.param
    set_encode_plan(0)


def _encode_bytes(tk, texts, **kw):
    """texts: list of equal-length bytes -> list of np.uint32 arrays."""
    n = len(texts[0])
    t = torch.from_numpy(np.frombuffer(b"".join(texts), dtype=np.uint8).reshape(len(texts), n).copy()).cuda()
    ids, counts = tk.encode_bytes(t, **kw)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    return [ids[b, : min(counts[b], ids.shape[1])].astype(np.uint32) for b in range(len(texts))], counts


# ---- quantiser -----------------------------------------------------------------------------
def test_quantiser_vs_reference_golden(dev):
    from ecg_byte_amd.tokenizer import quantize
    z = np.load(os.path.join(GOLDEN, "quantize_ref.npz"))
    for i in range(int(z["n_cases"])):
        p1, p99 = map(float, z[f"p_{i}"])
        pc = {"percentile_1": p1, "percentile_99": p99}
        x = torch.from_numpy(z[f"x_{i}"]).cuda()
        sym = quantize(x, pc).cpu().numpy()                                   # threshold kernel
        assert np.array_equal(sym, z[f"sym_{i}"]), f"case {i}: staircase kernel"
        clipped, sym2 = quantize(x, pc, want_clipped=True)                    # literal-division kernel
        assert np.array_equal(sym2.cpu().numpy(), z[f"sym_{i}"]), f"case {i}: division kernel"
        assert np.array_equal(clipped.cpu().numpy(), z[f"clipped_{i}"]), f"case {i}: clipped"


def test_quantiser_edges_nan_unaligned_tail(dev):
    from ecg_byte_amd.tokenizer import quantize
    rng = np.random.default_rng(0)
    pc = {"percentile_1": -0.3, "percentile_99": 0.9}
    for n in (1, 2, 3, 5, 255, 1027, 100003):
        x = rng.normal(0.3, 0.8, size=n)
        x[rng.integers(0, n, size=max(1, n // 50))] = np.nan
        got = quantize(torch.from_numpy(x).cuda(), pc).cpu().numpy()
        assert np.array_equal(got, O.quantize(x, pc["percentile_1"], pc["percentile_99"])), n   # NaN -> 0
    # degenerate percentiles (scale <= 0): literal-division kernel, same as the oracle
    bad = {"percentile_1": 5.0, "percentile_99": -10.0}
    x = rng.normal(0, 10, size=1000)
    got = quantize(torch.from_numpy(x).cuda(), bad).cpu().numpy()
    assert np.array_equal(got, O.quantize(x, 5.0, -10.0))
    # tokenizer_utils mirror returns the reference's shapes/dtypes
    from ecg_byte_amd import tokenizer_utils as tu
    c, s = tu.normalize_all(np.array([[0.0, 1.5, -2.0, 2.0]]), {"percentile_1": -1, "percentile_99": 1})
    assert s.dtype.kind == "U" and "".join(s[0]) == "mzaz" and c.dtype == np.float64


# ---- encoder: known-answer vectors through the drop-in module ---------------------------------
def test_rust_bpe_encode_text_known_answers(dev, plan):
    from ecg_byte_amd import rust_bpe
    assert rust_bpe.encode_text("abc", [([98, 99], 256), ([97, 98], 257)]) == [257, 99]
    m = [([97, 97], 256)]
    assert rust_bpe.encode_text("aaaa", m) == [256, 256]
    assert rust_bpe.encode_text("aaa", m) == [256, 97]
    assert rust_bpe.encode_text("abc", [([97, 98, 99], 256), ([97, 98, 99], 257)]) == [257]
    m = [([97, 98, 99, 100], 256)]
    assert rust_bpe.encode_text("abcd", m) == [256]
    assert rust_bpe.encode_text("abcx", m) == [97, 98, 99, 120]
    assert rust_bpe.encode_text("", m) == []
    assert rust_bpe.encode_text("Hello, wörld", []) == list("Hello, wörld".encode("utf-8"))
    assert rust_bpe.encode_text("a", [([97], 300)]) == [300]          # a length-1 expansion overrides the byte
    with pytest.raises(TypeError):
        rust_bpe.encode_text(123, m)


# ---- encoder vs oracle on seeded inputs -------------------------------------------------------
@pytest.mark.parametrize("tag,L", [("c1", 1000), ("c2", 5000)])
@pytest.mark.parametrize("B", [1, 3, 64])
def test_quantize_encode_fixture_tokenizers(dev, plan, tag, L, B):
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer
    vocab, merges, pc = load_tokenizer(tag)
    tk = HipTokenizer(merges)
    x = synth.synth_ecg(B, L, seed=0)
    ids, counts = tk.quantize_encode(torch.from_numpy(x).cuda(), pc)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    ref = oracle_batch(O.Trie(merges), x, pc)
    z = np.load(os.path.join(GOLDEN, "encode_oracle.npz"))
    for b in range(B):
        assert counts[b] == ref[b].size
        assert np.array_equal(ids[b, : counts[b]].astype(np.uint32), ref[b]), f"record {b}"
        if b < 3:
            assert np.array_equal(ids[b, : counts[b]], z[f"{tag}_ids_{b}"])   # committed anchor


def test_fused_staging_on_the_quantiser_golden_inputs(dev, plan):
    """The encode kernels classify samples while staging them (float fast path, exact staircase for a group with an
    ambiguous sample): run them on the reference-generated quantiser inputs, which sit +-4 ulp around every bin edge."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    _, merges, _ = load_tokenizer("c1")
    tk = HipTokenizer(merges)
    trie = O.Trie(merges)
    z = np.load(os.path.join(GOLDEN, "quantize_ref.npz"))
    for i in range(int(z["n_cases"])):
        p1, p99 = map(float, z[f"p_{i}"])
        x = np.ascontiguousarray(z[f"x_{i}"].reshape(1, -1))
        for n in (x.shape[1], x.shape[1] - 3):            # whole groups of four, and a ragged tail
            if n <= 0:
                continue
            xs = np.ascontiguousarray(np.tile(x[:, :n], (5, 1)))          # 5 records: also the wave-per-record kernel's rows
            ids, counts = tk.quantize_encode(torch.from_numpy(xs).cuda(), {"percentile_1": p1, "percentile_99": p99})
            ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
            want = trie.encode_bytes(O.symbols_to_text(z[f"sym_{i}"].reshape(-1)[:n])).astype(np.uint32)
            for b in range(5):
                assert counts[b] == want.size, f"case {i} n {n}"
                assert np.array_equal(ids[b, : counts[b]].astype(np.uint32), want), f"case {i} n {n}"


def test_fused_staging_with_a_huge_offset_takes_the_exact_route(dev, plan):
    """Percentiles 1e9 away from zero: (x - a) * scale cannot be formed as one fma accurately enough, the handle must fall
    back to the literal operation sequence -- the ids still equal the oracle's."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    _, merges, _ = load_tokenizer("c1")
    tk = HipTokenizer(merges)
    trie = O.Trie(merges)
    rng = np.random.default_rng(5)
    pc = {"percentile_1": 1e9, "percentile_99": 1e9 + 1.5}
    x = 1e9 + rng.uniform(-0.7, 2.2, size=(4, 12, 250))
    ids, counts = tk.quantize_encode(torch.from_numpy(x).cuda(), pc)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    for b in range(4):
        want = trie.quantize_encode(x[b], pc["percentile_1"], pc["percentile_99"])
        assert counts[b] == want.size and np.array_equal(ids[b, : counts[b]].astype(np.uint32), want)


def test_one_id_on_expansions_of_different_lengths(dev, plan):
    """The id -> length table of the wave-per-record kernel cannot describe such a vocabulary: the handle must route
    around it (lib.rs keeps whatever id the merges list gives a byte string)."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    merges = [([97, 98], 300), ([97, 98, 99], 300), ([98, 98], 301), ([97, 97, 97, 97], 301), ([99], 97)]
    tk = HipTokenizer(merges)
    rng = np.random.default_rng(0)
    texts = [bytes(rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=700)) for _ in range(6)]
    got, counts = _encode_bytes(tk, texts)
    for t, g in zip(texts, got):
        assert np.array_equal(g, np.asarray(O.encode_text(t, merges), dtype=np.uint32))


def test_four_streams_per_workgroup_with_ragged_tail(dev):
    """batch >= 2 x CUs takes the wave-per-stream kernel; 1030 records leave a partly filled last workgroup."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer
    _, merges, pc = load_tokenizer("c1")
    tk = HipTokenizer(merges)
    base = synth.synth_ecg(103, 1000, seed=4)
    x = np.concatenate([base] * 10)                                      # 1030 records
    ids, counts = tk.quantize_encode(torch.from_numpy(x).cuda(), pc)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    ref = oracle_batch(O.Trie(merges), base, pc)
    for b in range(x.shape[0]):
        r = ref[b % 103]
        assert counts[b] == r.size and np.array_equal(ids[b, : r.size].astype(np.uint32), r), b


@pytest.mark.parametrize("n", [1, 2, 31, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8193, 32767, 32768, 32769, 65537, 200001])
def test_stream_lengths_across_chunk_and_segment_boundaries(dev, plan, n):
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng(n)
    merges = random_merges(rng, 300, alphabet=b"abcd", max_len=9)
    tk = HipTokenizer(merges)
    texts = [bytes(rng.choice(np.frombuffer(b"aaabbcd", dtype=np.uint8), size=n)) for _ in range(3)]
    got, counts = _encode_bytes(tk, texts)
    for b, t in enumerate(texts):
        ref = O.encode_text(t, merges)
        assert counts[b] == len(ref) and np.array_equal(got[b], ref), (n, b)


def test_empty_batch_and_empty_streams(dev, plan):
    from ecg_byte_amd.tokenizer import HipTokenizer
    tk = HipTokenizer([([97, 98], 256)])
    ids, counts = tk.encode_bytes(torch.zeros((3, 0), dtype=torch.uint8, device="cuda"))
    assert counts.cpu().tolist() == [0, 0, 0]
    ids, counts = tk.quantize_encode(torch.zeros((0, 12, 10), dtype=torch.float64, device="cuda"),
                                     {"percentile_1": 0.0, "percentile_99": 1.0})
    assert ids.shape[0] == 0 and counts.shape[0] == 0


@pytest.mark.parametrize("seed", range(5))
def test_random_merges_duplicates_interior_nodes_extra_bytes(dev, plan, seed):
    """Non-prefix-closed vocabularies, duplicate expansions (last wins), bytes outside a..z in the
    merges (extra symbol classes) and bytes that occur in no merge at all."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng(100 + seed)
    alphabet = [b"abc", b"abcdefgh", b"xyz.,", b"ab\x00\xff", b"mnopqr \n"][seed]
    merges = random_merges(rng, int(rng.integers(20, 600)), alphabet=alphabet, max_len=14, dup_frac=0.15)
    tk = HipTokenizer(merges)
    pool = np.frombuffer(alphabet + b"~Q", dtype=np.uint8)
    for n in (7, 1000, 5003):
        texts = [bytes(rng.choice(pool, size=n)) for _ in range(4)]
        got, counts = _encode_bytes(tk, texts)
        for b, t in enumerate(texts):
            ref = O.encode_text(t, merges)
            assert counts[b] == len(ref) and np.array_equal(got[b], ref), (seed, n, b)


def test_chains_that_never_resynchronise(dev, plan):
    """Adversarial for the speculative chunk parse: with the single token 'ab' over 'ababab...'
    a parse started at an odd offset never meets the true chain, and long same-symbol runs with
    power-of-two tokens make the true chain jump over whole chunks.  The fixed-point stitch
    must still return the sequential answer."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    cases = []
    cases.append(([([97, 98], 256)], b"b" + b"ab" * 40000))             # every chunk starts mis-phased
    cases.append(([([97, 98], 256)], b"ab" * 40000 + b"a"))
    runs = [([97] * (1 << k), 256 + k - 1) for k in range(1, 11)]        # a^2 .. a^1024
    cases.append((runs, b"a" * 70001))
    cases.append((runs, (b"a" * 1000 + b"b") * 90))
    three = [([97, 98, 99], 256), ([98, 99, 97], 257), ([99, 97, 98], 258)]
    cases.append((three, b"c" + b"abc" * 30000))
    for merges, text in cases:
        tk = HipTokenizer(merges)
        got, counts = _encode_bytes(tk, [text])
        ref = O.encode_text(text, merges)
        assert counts[0] == len(ref) and np.array_equal(got[0], ref)


def test_ids_stride_truncates_but_counts_full_and_prefix_stable(dev, plan):
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer
    _, merges, pc = load_tokenizer("c2")
    tk = HipTokenizer(merges)
    x = synth.synth_ecg(5, 5000, seed=9)
    xd = torch.from_numpy(x).cuda()
    full, counts = tk.quantize_encode(xd, pc)
    cut, counts2 = tk.quantize_encode(xd, pc, ids_stride=1020)
    assert torch.equal(counts, counts2) and cut.shape == (5, 1020)
    assert torch.equal(cut, full[:, :1020])


# ---- full-size properties (BASELINE configs[1]: 12x5000, vocab 4k) ----------------------------
def test_full_size_properties(dev):
    """4096 records of 12x5000: every record's token lengths sum to 60000 (a checksum of the whole
    decode), ids are in range, and a seeded sample of records equals the oracle bit for bit."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer
    vocab, merges, pc = load_tokenizer("c2")
    tk = HipTokenizer(merges)
    B, L = 4096, 5000
    base = synth.synth_ecg(256, L, seed=21)
    rng = np.random.default_rng(3)
    gains = rng.uniform(0.8, 1.2, size=(B // 256, 1, 1, 1))
    x = (base[None] * gains).reshape(B, 12, L)                           # 4096 distinct records
    xd = torch.from_numpy(x).cuda()
    ids, counts = tk.quantize_encode(xd, pc)
    tok_len = np.zeros(256 + len(merges), dtype=np.int64)
    tok_len[:256] = 1
    for seq, tid in merges:
        tok_len[tid] = len(seq)
    tl = torch.from_numpy(tok_len).cuda()
    col = torch.arange(ids.shape[1], device="cuda")[None, :]
    valid = col < counts[:, None]
    idl = ids.long().clamp_(0, tok_len.size - 1)
    assert bool(((ids >= 0) & (ids < tok_len.size) | ~valid).all())
    sums = (tl[idl] * valid).sum(dim=1)
    assert bool((sums == 12 * L).all()), "token lengths do not tile the symbol stream"
    trie = O.Trie(merges)
    ids_h, counts_h = ids.cpu().numpy(), counts.cpu().numpy()
    for b in rng.choice(B, size=48, replace=False):
        ref = trie.quantize_encode(x[b], pc["percentile_1"], pc["percentile_99"])
        assert counts_h[b] == ref.size and np.array_equal(ids_h[b, : ref.size].astype(np.uint32), ref), b
        if b % 7 == 0:   # reference's own check: decode(encode(x)) == x (train_tokenizer.py:58-60)
            text = O.symbols_to_text(O.quantize(x[b], pc["percentile_1"], pc["percentile_99"])).decode()
            assert O.decode_text(ids_h[b, : counts_h[b]].tolist(), vocab) == text
    # idempotence: same input, same output
    ids2, counts2 = tk.quantize_encode(xd, pc)
    assert torch.equal(counts, counts2) and bool(((ids == ids2) | ~valid).all())


# ---- sequence assembly ------------------------------------------------------------------------
def test_assemble_vs_reference_golden(dev):
    from ecg_byte_amd.data_loader import BatchAssembler
    with open(os.path.join(GOLDEN, "assemble_ref.json")) as f:
        cases = json.load(f)
    for c in cases:
        sp = c["special"]
        sig = np.asarray(c["sig"], dtype=np.int64)
        # identity-shifted LUT: tokenizer id k -> LLM id k + 128260 covers the golden's id range
        lut = np.arange(0, 4000, dtype=np.int32) + 128260
        asm = BatchAssembler([([97, 98], 256)], lut, sp["<pad>"], sp["<bos>"], sp["<eos>"],
                             sp["<sig_start>"], sp["<sig_end>"], c["pad_to_max"])
        n = max(1, sig.size)
        ids = torch.zeros((1, n), dtype=torch.int32, device="cuda")
        ids[0, : sig.size] = torch.from_numpy((sig - 128260).astype(np.int32)).cuda()
        counts = torch.tensor([sig.size], dtype=torch.int32, device="cuda")
        r = asm.assemble(ids, counts, [c["q"]], [c["a"]])
        for k in ("tokenized_signal", "attn_mask", "quantized_signal_ids_input", "position_ids"):
            assert np.array_equal(r[k][0].cpu().numpy(), np.asarray(c[k])), (c["pad_to_max"], k)
        inf = asm.assemble(ids, counts, [c["q"]], inference=True)
        m = int(inf["lengths"][0])
        assert np.array_equal(inf["tokenized_signal"][0, :m].cpu().numpy(), np.asarray(c["inference_tokenized_signal"]))
        assert np.array_equal(inf["attn_mask"][0, :m].cpu().numpy(), np.asarray(c["inference_attn_mask"]))


def test_batch_assembler_end_to_end_vs_oracle_pipeline(dev):
    """signal -> quantise -> encode -> LUT -> rows, batch of 33, against the oracle pipeline
    (quantize_encode oracle + restated _prepare_training), incl. pad ids appearing inside Q."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.data_loader import BatchAssembler
    vocab, merges, pc = load_tokenizer("c2")
    rng = np.random.default_rng(17)
    keys = list(vocab.keys())
    rng.shuffle(keys)                                                     # pickled-dict order is arbitrary
    lut = np.zeros(max(keys) + 1, dtype=np.int32)
    lut[keys] = 128256 + np.arange(len(keys))
    pad, bos, eos, s0, s1 = 132014, 128000, 128001, 132012, 132013
    B, L, P = 33, 5000, 1020
    x = synth.synth_ecg(B, L, seed=33)
    x[5, :, 600:] = 0.0                                                   # a highly compressible record -> padded row
    x[6] = 0.0
    qs = [rng.integers(1000, 100000, size=int(rng.integers(0, 25))).tolist() for _ in range(B)]
    ans = [rng.integers(1000, 100000, size=int(rng.integers(0, 33))).tolist() for _ in range(B)]
    qs[2][3:5] = [pad, pad]                                               # mask is by VALUE (data_loader.py:21-22)
    asm = BatchAssembler(merges, lut, pad, bos, eos, s0, s1, P)
    out = asm(torch.from_numpy(x).cuda(), pc, qs, ans)
    trie = O.Trie(merges)
    n_padded = 0
    for b in range(B):
        sig = lut[trie.quantize_encode(x[b], pc["percentile_1"], pc["percentile_99"])]
        r = OA.prepare_training(sig.tolist(), qs[b], ans[b], pad, bos, eos, s0, s1, P)
        n_padded += int(r["tokenized_signal"][0] == pad)
        for k in r:
            assert np.array_equal(out[k][b].cpu().numpy(), r[k]), (b, k)
        assert out["attn_mask"].dtype == torch.float32 and out["position_ids"].dtype == torch.int64
    assert n_padded >= 2
    with pytest.raises(AssertionError):
        asm.assemble(*asm.encode(torch.from_numpy(x[:1]).cuda(), pc, max_tokens=P), [list(range(1000))], [list(range(30))])


def test_train_tokenizer_cli_end_to_end(dev, tmp_path):
    """The reference's train_tokenizer.py flow on the reference's on-disk formats: .npy records,
    a sampled-files list, a pickled-dict percentiles .npy -> tokenizer .pkl + round-trip check."""
    from ecg_byte_amd import synth, train_tokenizer
    from ecg_byte_amd import tokenizer_utils as tu
    pc = synth.synth_percentiles(250, seed=0, n_samples=20000)
    np.save(tmp_path / "pc.npy", pc)                                 # preprocess_utils.py:208-210
    paths = []
    x = synth.synth_ecg(12, 250, seed=8)
    for i in range(12):
        p = tmp_path / f"ecg_{i}_0.npy"
        np.save(p, x[i])
        paths.append(str(p))
    (tmp_path / "sampled.txt").write_text("\n".join(paths) + "\n")
    out = tmp_path / "tokenizer_50.pkl"
    args = train_tokenizer.get_args(["--train", "--num_merges", "50", "--sampled_files", str(tmp_path / "sampled.txt"),
                                     "--percentiles", str(tmp_path / "pc.npy"), "--check_ecg", paths[3], "--out", str(out)])
    assert train_tokenizer.main(args) is True
    vocab, merges = tu.load_vocab_and_merges(str(out))
    corpus = tu.process_large_file(str(tmp_path / "sampled.txt"), pc, 2)
    assert corpus == "".join(O.symbols_to_text(O.quantize(x[i], pc["percentile_1"], pc["percentile_99"])).decode() for i in range(12))
    assert (vocab, merges) == O.byte_pair_encoding(corpus, 50, fast=False)[1:]


def test_trie_larger_than_lds(dev, plan):
    """~30 000 trie nodes (240 KB) do not fit the 160 KB of LDS next to the per-wave buffers: nodes beyond `n_lds` are read
    through L2 by both kernels (ALL_LDS = false instantiations)."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng(77)
    merges = random_merges(rng, 6500, alphabet=b"abcdefghijklmnopqrstuvwxyz", max_len=12, dup_frac=0.02)
    tk = HipTokenizer(merges)
    assert tk.n_nodes > 14000
    pool = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
    texts = [bytes(rng.choice(pool, size=20011)) for _ in range(3)]
    # plant expansions so that deep (non-LDS) nodes are actually reached
    for t in range(3):
        buf = bytearray(texts[t])
        for k in range(300):
            seq = bytes(merges[int(rng.integers(len(merges)))][0])
            at = int(rng.integers(0, len(buf) - len(seq)))
            buf[at:at + len(seq)] = seq
        texts[t] = bytes(buf)
    got, counts = _encode_bytes(tk, texts)
    for b, t in enumerate(texts):
        ref = O.encode_text(t, merges)
        assert counts[b] == len(ref) and np.array_equal(got[b], ref), b


def test_run_chains_with_branches_and_sparse_tokens(dev, plan):
    """The run step: same-class chains up to 200 deep where only some depths carry a token (a^k for k in a sparse set),
    branches hanging off the middle of a chain (a^k b, a^k c c), two interleaved run alphabets, runs longer than the
    chain, runs that end exactly on / one short of / one past a token, and the 32-symbol cap of one step."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng(5)
    merges, tid = [], 256
    for k in (2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 129, 200):
        merges.append(([97] * k, tid)); tid += 1
        if k in (3, 8, 33, 100):
            merges.append(([97] * k + [98], tid)); tid += 1
            merges.append(([97] * k + [99, 99], tid)); tid += 1
    for k in (2, 5, 6, 40, 41, 150):
        merges.append(([98] * k, tid)); tid += 1
    merges.append(([98, 97], tid)); tid += 1
    merges.append(([99, 98, 98, 98, 97], tid)); tid += 1
    tk = HipTokenizer(merges)
    texts = []
    for _ in range(6):
        parts = []
        while sum(len(p) for p in parts) < 30000:
            ch = bytes([int(rng.choice([97, 97, 97, 98, 98, 99]))])
            parts.append(ch * int(rng.choice([1, 1, 2, 3, 7, 8, 9, 31, 32, 33, 34, 63, 64, 65, 99, 100, 101, 128, 199, 200, 201, 450])))
        texts.append(b"".join(parts)[:30000])
    got, counts = _encode_bytes(tk, texts)
    for b, t in enumerate(texts):
        ref = O.encode_text(t, merges)
        assert counts[b] == len(ref) and np.array_equal(got[b], ref), b


# ---- encode_long_kernel (round 5): one wave per record, one lane per long chunk of run-length entries ---------------------------------------
def _signal_for_symbols(sym, pc):
    """float64 samples in the middle of the bins of `sym` (alphabet indices 0..25) under normalize_all's map (tokenizer_utils.py:14-19)."""
    a = pc["percentile_1"] - 0.5
    d = ((pc["percentile_99"] + 0.5) - a) + 1e-6
    return a + (np.asarray(sym, dtype=np.float64) + 0.5) * d / 26.0


LONG_PLAN = 4      # 4: encode_long_kernel; the tests below run a second time under 6 (encode_pipe_kernel: stagers and walkers in one workgroup)


def _check_long(tk, merges, syms, pc, **kw):
    """syms: (B, n) alphabet indices; quantise + encode on the lane-per-chunk kernel vs lib.rs's restatement on the symbol text."""
    from ecg_byte_amd.tokenizer import set_encode_plan
    x = _signal_for_symbols(syms, pc)
    set_encode_plan(LONG_PLAN)
    try:
        ids, counts = tk.quantize_encode(torch.from_numpy(np.ascontiguousarray(x)).cuda(), pc, **kw)
    finally:
        set_encode_plan(0)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    for b in range(syms.shape[0]):
        ref = np.asarray(O.encode_text(bytes((np.asarray(syms[b]) + 97).astype(np.uint8)), merges), dtype=np.uint32)
        assert counts[b] == ref.size, (b, counts[b], ref.size)
        k = min(ref.size, ids.shape[1])
        assert np.array_equal(ids[b, :k].astype(np.uint32), ref[:k]), f"record {b}"


PC = {"percentile_1": -1.0, "percentile_99": 1.0}


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 4097, 12000, 60000, 65534, 65535])
def test_lane_per_chunk_record_lengths(dev, n):
    """Odd and even lengths (the vector and the scalar staging), fewer runs than lanes, tile and block boundaries, the 16-bit position limit."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng(n)
    merges = random_merges(rng, 300, alphabet=b"abcdefgh", max_len=9)
    tk = HipTokenizer(merges)
    # runs of random lengths (1 .. 40) over eight classes, three records of different texture
    recs = []
    for mean_run in (1.2, 4.0, 25.0):
        s = np.repeat(rng.integers(0, 8, size=n), rng.geometric(1.0 / mean_run, size=n))[:n]
        recs.append(s)
    _check_long(tk, merges, np.stack(recs), PC)


def test_lane_per_chunk_parses_that_never_meet_and_long_runs(dev):
    """'ab' over 'babab...' (a parse entered at the wrong phase never meets the owner's: a lane's list overflows and the wave redoes the record with one lane);
    a^(2^k) tokens over runs of 1 000 (back-ups of hundreds of symbols across the forced 256-symbol run boundaries); 'abc' rotations."""
    from ecg_byte_amd.tokenizer import HipTokenizer
    n = 60000
    ab = np.tile([1, 0], n // 2 + 1)
    cases = [([([97, 98], 256)], np.stack([ab[:n], ab[1:n + 1]])),
             ([([97] * (1 << k), 256 + k - 1) for k in range(1, 8)], np.stack([np.zeros(n, dtype=np.int64), np.tile(np.r_[np.zeros(1000, dtype=np.int64), 1], 60)[:n]])),
             ([([97, 98, 99], 256), ([98, 99, 97], 257), ([99, 97, 98], 258)], np.stack([np.tile([2, 0, 1], n // 3 + 1)[:n], np.tile([0, 1, 2], n // 3 + 1)[:n]]))]
    for merges, syms in cases:
        _check_long(HipTokenizer(merges), merges, syms, PC)


@pytest.mark.parametrize("tag,L,B", [("c1", 1000, 5), ("c2", 5000, 3), ("c2", 5000, 300)])
def test_lane_per_chunk_fixture_tokenizers(dev, tag, L, B):
    """The committed tokenizers on synthetic records; 300 records = several records per resident wave slot (scratch reuse)."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
    vocab, merges, pc = load_tokenizer(tag)
    tk = HipTokenizer(merges)
    x = synth.synth_ecg(min(B, 64), L, seed=4)
    if B > 64:
        x = np.concatenate([x] * ((B + 63) // 64))[:B]
    set_encode_plan(LONG_PLAN)
    try:
        ids, counts = tk.quantize_encode(torch.from_numpy(x).cuda(), pc)
        cut, counts2 = tk.quantize_encode(torch.from_numpy(x).cuda(), pc, ids_stride=1020)
    finally:
        set_encode_plan(0)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    ref = oracle_batch(O.Trie(merges), x[: min(B, 64)], pc)
    for b in range(B):
        r = ref[b % 64]
        assert counts[b] == r.size and np.array_equal(ids[b, : r.size].astype(np.uint32), r), f"record {b}"
    assert np.array_equal(counts2.cpu().numpy(), counts) and np.array_equal(cut.cpu().numpy(), ids[:, :1020])


def test_lane_per_chunk_equals_segment_kernels_on_the_bench_batch(dev):
    """4 096 records of 12 x 5000 (BASELINE configs[1]): the lane-per-chunk kernel (plan 4) and the segment kernel of the automatic plan give the same ids and counts."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
    _, merges, pc = load_tokenizer("c2")
    tk = HipTokenizer(merges)
    base = synth.synth_ecg(256, 5000, seed=33)
    gains = np.random.default_rng(5).uniform(0.7, 1.3, size=(16, 1, 1, 1))
    xd = torch.from_numpy((base[None] * gains).reshape(4096, 12, 5000)).cuda()
    ids0, counts0 = tk.quantize_encode(xd, pc)
    set_encode_plan(LONG_PLAN)
    try:
        ids, counts = tk.quantize_encode(xd, pc)
    finally:
        set_encode_plan(0)
    assert torch.equal(counts, counts0)
    valid = torch.arange(ids.shape[1], device="cuda")[None, :] < counts[:, None]
    assert bool(((ids == ids0) | ~valid).all())


# ---- the general form (round 5): merges the packed trie cannot hold ---------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["256-byte-values", "wide-ids", "many-nodes"])
def test_general_form_byte_level_merges_vs_oracle(dev, kind):
    """Random byte-level merges over all 256 byte values / token ids above 65 535 / a trie of more than 65 535 nodes: `encode_text` through the drop-in module
    and `encode_bytes` on batches, against lib.rs's restatement (lib.rs:149-193)."""
    from ecg_byte_amd import rust_bpe
    from ecg_byte_amd.tokenizer import HipTokenizer
    rng = np.random.default_rng({"256-byte-values": 1, "wide-ids": 2, "many-nodes": 3}[kind])
    if kind == "256-byte-values":
        merges = random_merges(rng, 800, alphabet=bytes(range(256)), max_len=6)
        texts = [bytes(rng.integers(0, 256, size=5000).astype(np.uint8)) for _ in range(5)]
        # make the merges hit: splice expansions into the texts
        texts = [b"".join(bytes(merges[int(k)][0]) for k in rng.integers(0, 800, size=900))[:5000].ljust(5000, b"\x00") for _ in range(5)] + texts[:2]
    elif kind == "wide-ids":
        merges = [(seq, 70000 + 3 * i) for i, (seq, _) in enumerate(random_merges(rng, 300, alphabet=b"abcdef", max_len=8))]
        texts = [bytes(rng.choice(list(b"abcdef"), size=4000).astype(np.uint8)) for _ in range(4)]
    else:
        merges = [(rng.integers(97, 123, size=6).tolist(), 256 + i) for i in range(24000)]
        texts = [bytes(rng.integers(97, 123, size=3000).astype(np.uint8)) for _ in range(3)]
        texts.append(b"".join(bytes(merges[int(k)][0]) for k in rng.integers(0, 24000, size=500)))
        texts = [t[:3000].ljust(3000, b"a") for t in texts]
    tk = HipTokenizer(merges)
    got, counts = _encode_bytes(tk, texts)
    for b, t in enumerate(texts):
        ref = O.encode_text(t, merges)
        assert counts[b] == len(ref) and np.array_equal(got[b], np.asarray(ref, dtype=np.uint32)), (kind, b)
    if kind != "256-byte-values":                                           # the module takes `str`: ASCII texts
        t = texts[0].decode("ascii")
        assert rust_bpe.encode_text(t, merges) == [int(v) for v in O.encode_text(texts[0], merges)]


def test_general_form_quantize_encode(dev):
    """The fused entry point with a general-form handle: quantise into the scratch buffer, then the general walk."""
    from ecg_byte_amd import synth
    from ecg_byte_amd.tokenizer import HipTokenizer
    _, merges, pc = load_tokenizer("c1")
    wide = [(seq, 100000 + tid) for seq, tid in merges]                     # the C1 merges with ids that need 32 bits
    tk = HipTokenizer(wide)
    x = synth.synth_ecg(6, 1000, seed=2)
    ids, counts = tk.quantize_encode(torch.from_numpy(x).cuda(), pc)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    for b in range(6):
        text = O.symbols_to_text(O.quantize(x[b], pc["percentile_1"], pc["percentile_99"]))
        ref = np.asarray(O.encode_text(text, wide), dtype=np.uint32)
        assert counts[b] == ref.size and np.array_equal(ids[b, : ref.size].astype(np.uint32), ref), b


@pytest.fixture
def pipe_plan():
    global LONG_PLAN
    LONG_PLAN = 6
    yield
    LONG_PLAN = 4


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 4097, 12000, 60000, 65534, 65535])
def test_pipelined_stage_walk_record_lengths(dev, pipe_plan, n):
    test_lane_per_chunk_record_lengths(dev, n)


def test_pipelined_stage_walk_parses_that_never_meet_and_long_runs(dev, pipe_plan):
    test_lane_per_chunk_parses_that_never_meet_and_long_runs(dev)


@pytest.mark.parametrize("tag,L,B", [("c1", 1000, 5), ("c2", 5000, 3), ("c2", 5000, 300), ("c2", 5000, 9000)])
def test_pipelined_stage_walk_fixture_tokenizers(dev, pipe_plan, tag, L, B):
    """9 000 records: more than two rounds of sixteen records per workgroup (slots reused: the stager waits for the walker)."""
    test_lane_per_chunk_fixture_tokenizers(dev, tag, L, B)


def test_pipelined_stage_walk_equals_segment_kernels_on_the_bench_batch(dev, pipe_plan):
    test_lane_per_chunk_equals_segment_kernels_on_the_bench_batch(dev)
