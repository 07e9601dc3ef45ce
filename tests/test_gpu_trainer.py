"""GPU tests of the HIP BPE trainer against the CPU oracle (literal lib.rs loop and its
fast equivalent), which share the DEFINED tie-break (smallest (left,right)); plus the
properties the reference guarantees whatever its tie-break."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(autouse=True, params=[0, 1, 2, 3], ids=["slotted16", "slotted32", "two_pass", "slotted16_one_launch"])
def train_form(request):
    """Every test of the file under the merge steps of ecgb_bpe_train_hip (ecgb_set_bpe_train_form / _fused): ranges in fixed slots with one pass per merge and 16-bit ids
    (the default), the same with 32-bit ids, round 4's count pass + rewrite, and the default form with the next merge's row maxima inside the merge's launch (round 6: ONE
    launch per merge; built, merge-for-merge the same, 2 % slower -- not the default).  The sharded form has its own kernels: once."""
    from ecg_byte_amd import trainer
    name = request.node.name
    if request.param and ("sharded" in name or "kept_row_maxima" in name):
        pytest.skip("does not go through ecgb_bpe_train_hip's merge step")
    trainer.set_train_form(0 if request.param == 3 else request.param)
    trainer.set_train_fused(request.param == 3)
    yield request.param
    trainer.set_train_form(0)
    trainer.set_train_fused(False)


def _train(text, num_merges):
    from ecg_byte_amd import rust_bpe
    return rust_bpe.byte_pair_encoding(text, num_merges, 1)


def test_trainer_known_answers():
    ids, vocab, merges = _train("aaabdaaabac", 2)
    assert merges == [([97, 97], 256), ([97, 98], 257)]              # step 1: tie -> smallest pair
    assert ids == [256, 257, 100, 256, 257, 97, 99]
    assert vocab[256] == "aa" and vocab[257] == "ab"
    assert _train("ab", 5)[0] == [256] and len(_train("ab", 5)[2]) == 1   # stops when no pair is left
    assert _train("", 3) == ([], {i: O.byte_to_string(i) for i in range(256)}, [])
    assert _train("a", 3)[0] == [97]
    assert _train("aaaa", 1)[0] == [256, 256]
    assert _train("aaa", 1)[0] == [256, 97]                              # merge([a,a,a]) = [X,a]


@pytest.mark.parametrize("seed", range(6))
def test_trainer_equals_oracle_small(seed):
    rng = np.random.default_rng(seed)
    for _ in range(6):
        n = int(rng.integers(0, 3000))
        k = int(rng.choice([1, 2, 3, 5, 26]))
        text = bytes(rng.integers(97, 97 + k, size=n).astype(np.uint8))
        nm = int(rng.integers(0, 70))
        got = _train(text, nm)
        assert got == O.byte_pair_encoding(text, nm, fast=False), (seed, n, k, nm)


@pytest.mark.parametrize("n,alphabet", [(4095, b"a"), (4096, b"a"), (4097, b"a"), (20000, b"ab"), (70001, b"aab"),
                                        (16 * 4096 + 7, b"a"), (9000, b"abab")])
def test_trainer_runs_across_thread_and_tile_boundaries(n, alphabet):
    """Long same-symbol runs (merge of a symbol with itself) across 16-id spans and 4096-id tiles."""
    rng = np.random.default_rng(n)
    text = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n))
    if alphabet == b"a":
        text = b"b" * (n % 5) + text      # shift the run against the tile grid
    assert _train(text, 14) == O.byte_pair_encoding(text, 14, fast=True)


def test_trainer_on_ecg_corpus_matches_fixture_prefix():
    """200 synthetic records (12x1000): the first 300 merges equal the oracle trainer's, the ids decode
    back to the text (train_tokenizer.py:58-60) and each chosen pair was a most-frequent pair."""
    from ecg_byte_amd import synth
    from helpers import load_tokenizer
    _, _, pc = load_tokenizer("c1")
    x = synth.synth_ecg(200, 1000, seed=1)
    text = O.symbols_to_text(O.quantize(x, pc["percentile_1"], pc["percentile_99"]))
    ids, vocab, merges = _train(text, 300)
    ref = O.byte_pair_encoding(text, 300, fast=True)
    assert merges == ref[2]
    assert ids == ref[0]
    assert O.decode_text(ids, vocab) == text.decode("ascii")
    assert all(m[1] == 256 + i for i, m in enumerate(merges))


def test_trained_tokenizer_feeds_the_encoder():
    """train -> encode round trip entirely on the device path."""
    from ecg_byte_amd import rust_bpe
    rng = np.random.default_rng(2)
    text = bytes(rng.choice(np.frombuffer(b"aaabbc", dtype=np.uint8), size=50000))
    ids, vocab, merges = _train(text, 200)
    enc = rust_bpe.encode_text(text.decode(), merges)
    assert "".join(vocab[i] for i in enc) == text.decode()
    assert enc == O.encode_text(text, merges)


@pytest.mark.parametrize("tag,L,nm", [("c1", 1000, 1000), ("c2", 5000, 4000)])
def test_trainer_reproduces_committed_tokenizers_at_full_size(tag, L, nm):
    """SURVEY.md §8d tokenizers: 2 000 synthetic records (seed 1) -> the committed (vocab, merges)
    pickles, which the ORACLE trainer produced.  Whole pipeline on the device: quantise -> train."""
    import bench
    from helpers import load_tokenizer
    from ecg_byte_amd import rust_bpe
    from ecg_byte_amd.tokenizer import quantize
    from ecg_byte_amd.trainer import bpe_train_device
    vocab, merges, pc = load_tokenizer(tag)
    x = bench.make_signals(2000, L, seed=1, start=0, workers=8)
    text = (quantize(torch.from_numpy(x).cuda(), pc).view(-1) + 97).contiguous()
    del x
    ids, n_ids, pairs, n_done = bpe_train_device(text, nm)
    assert int(n_done) == nm
    v2, m2 = rust_bpe.vocab_merges_from_pairs(pairs.cpu().tolist())
    assert m2 == merges and v2 == vocab
    # the final ids tile the corpus: token lengths sum to the text length
    lens = np.ones(256 + nm, dtype=np.int64)
    for seq, tid in merges:
        lens[tid] = len(seq)
    got = ids[: int(n_ids)].cpu().numpy()
    assert int(lens[got].sum()) == text.numel()


def test_trainer_past_2_pow_31_symbols():
    """A corpus longer than 2^31 symbols (the reference's is up to 6e9, tokenizer_utils.py:79-93): positions, tile offsets and pair
    counts are 64-bit.  The oracle cannot run at this size; checked through properties it guarantees: the first merge is the most
    frequent pair of the text (smallest pair among equals), every later merge's ids are 256 + i, and expanding the final ids by the
    merges gives the text back (train_tokenizer.py:58-60's round trip) -- on the device, at full length."""
    from ecg_byte_amd.trainer import bpe_train_device
    n = (1 << 31) + 12345
    g = torch.Generator(device="cuda").manual_seed(7)
    block = torch.randint(97, 101, (1 << 20,), device="cuda", generator=g, dtype=torch.uint8)
    block[1000:300000] = 109                                              # a long flat run inside every block: run-parity merges ('mm')
    text = block.repeat((n >> 20) + 1)[:n].contiguous()
    n_merges = 6
    ids, n_ids, pairs, n_done = bpe_train_device(text, n_merges)
    torch.cuda.synchronize()
    assert int(n_done) == n_merges
    m = int(n_ids)
    assert 0 < m < n
    pairs = pairs.cpu().tolist()
    # most frequent adjacent pair of the text, on the device in chunks (int64 keys of 2^31 positions do not fit one bincount comfortably)
    counts = torch.zeros(65536, dtype=torch.int64, device="cuda")
    step = 1 << 28
    for lo in range(0, n - 1, step):
        hi = min(n - 1, lo + step)
        key = text[lo:hi].to(torch.int32) * 256 + text[lo + 1: hi + 1].to(torch.int32)
        counts += torch.bincount(key, minlength=65536)
    best = int(torch.argmax(counts))                                      # argmax returns the first (= smallest) index among equals
    assert pairs[0] == [best // 256, best % 256], (pairs[0], best)
    # round trip: expand the ids merge by merge, newest first
    cur = ids[:m].to(torch.int32)
    for i in reversed(range(n_merges)):
        is_new = cur == 256 + i
        reps = 1 + is_new.to(torch.int64)
        out = torch.repeat_interleave(cur, reps)
        first = torch.cumsum(reps, 0) - reps                              # position of every element's first copy
        idx = first[is_new]
        out[idx] = pairs[i][0]
        out[idx + 1] = pairs[i][1]
        cur = out
        del out, reps, first, idx, is_new
    assert cur.numel() == n
    for lo in range(0, n, step):
        assert torch.equal(cur[lo: lo + step].to(torch.uint8), text[lo: lo + step]), lo


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_trainer_ranks_on_one_gpu(tmp_path, world):
    """SURVEY.md section 8e row 3 with the real kernels: the corpus cut into contiguous slices, one per rank (all on cuda:0 here, the
    exchange over gloo), merges and concatenated ids equal to the oracle trainer's on the whole text -- small alphabets (ties, self
    merges), cuts inside runs, empty slices, runs longer than a 4096-id tile, and a synthetic-ECG corpus of 240 000 symbols."""
    import os
    import pickle
    import subprocess
    import sys
    from helpers import load_tokenizer
    from ecg_byte_amd import synth
    rng = np.random.default_rng(5 + world)
    cases = []
    for trial in range(14):
        k = int(rng.choice([1, 2, 3, 5]))
        n = int(rng.integers(0, 9000))
        text = bytes(rng.integers(97, 97 + k, size=n).astype(np.uint8))
        if trial % 4 == 0:
            text = b"a" * int(rng.integers(0, 9000)) + text[: n // 3] + b"b" * int(rng.integers(0, 5000)) + b"a" * int(rng.integers(0, 6000))
        cuts = sorted(int(c) for c in rng.integers(0, len(text) + 1, size=world - 1))
        if trial % 6 == 0:
            cuts = [0] * (world - 1)
        cases.append((text, cuts, int(rng.integers(0, 60))))
    _, _, pc = load_tokenizer("c1")
    x = synth.synth_ecg(4, 5000, seed=9)
    sym = b"".join(O.symbols_to_text(O.quantize(x[b], pc["percentile_1"], pc["percentile_99"])) for b in range(4))
    cases.append((sym, [len(sym) * (r + 1) // world + 7 for r in range(world - 1)], 300))
    here = os.path.dirname(os.path.abspath(__file__))
    with open(tmp_path / "cases.pkl", "wb") as f:
        pickle.dump(cases, f)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29530 + world), os.path.join(here, "shard_worker.py"), str(tmp_path / "cases.pkl"), str(tmp_path / "out")]
    res = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    outs = [pickle.load(open(str(tmp_path / "out") + f".rank{r}", "rb")) for r in range(world)]
    for c, (text, cuts, nm) in enumerate(cases):
        want_ids, _, want_merges = O.byte_pair_encoding(text, nm, fast=True)
        for r in range(world):
            assert O.pairs_to_vocab_merges([tuple(p) for p in outs[r][c][1]])[1] == want_merges, (c, r, len(text), cuts, nm)
        assert sum((outs[r][c][0] for r in range(world)), []) == want_ids, (c, len(text), cuts, nm)


@pytest.mark.gpu
def test_kept_row_maxima_pick_the_pairs_a_full_scan_of_the_table_picks_past_16384_ids():
    """The one-rank trainer's arg-max keeps a maximum per table row and re-reads only the rows a merge can have moved (rowmax_kernel); the step-wise
    (sharded) form scans the live table every merge (argmax_partial_kernel).  Same corpus, one rank: the same 17 000 merges and the same ids -- past the 16 384
    ids one trip of rowmax_kernel covers, through merges of counts 1 with their ties, and down to a handful of ids left."""
    from ecg_byte_amd.trainer import HipShard, bpe_train_device
    rng = np.random.default_rng(17)
    n, nm = 120_000, 17_000
    # runs and repeats: a small alphabet with sticky symbols (l == r merges included), so that counts stay tied often
    sym = rng.integers(0, 40, size=n)
    stick = rng.random(n) < 0.35
    for i in range(1, n):
        if stick[i]:
            sym[i] = sym[i - 1]
    text = torch.from_numpy((sym + 40).astype(np.uint8)).cuda()
    ids_a, n_ids, pairs_a, n_done = bpe_train_device(text, nm)
    k = int(n_done)
    shard = HipShard(text, nm)
    summary, gathered = shard.new_words(8), shard.new_words(8)
    shard.begin(summary); gathered.copy_(summary)
    shard.count(gathered, 0, 1)
    for i in range(nm):
        shard.pick(i, summary); gathered.copy_(summary)
        shard.merge(i, gathered, 0, 1)
        shard.apply()
    ids_b, pairs_b = shard.finish()
    assert k == len(pairs_b) and k > 16_500, (k, len(pairs_b))
    assert pairs_a[:k].cpu().tolist() == [list(p) for p in pairs_b]
    assert ids_a[: int(n_ids)].cpu().tolist() == ids_b


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [1, 2, 3, 7, 50])
def test_few_workgroups_walk_ranges_of_many_tiles_to_the_same_merges(grid):
    """A workgroup of the count and rewrite passes walks a contiguous range of tiles and carries the output offset and the run parity across them; the workgroups
    chain their ranges' records.  With a handful of workgroups a small corpus already makes ranges of ten and more tiles -- runs of one symbol (merged with itself)
    across tile and range boundaries included: the oracle's merges and ids whatever the number of workgroups."""
    from ecg_byte_amd import trainer
    rng = np.random.default_rng(100 + grid)
    parts, total, n = [], 0, 45_000
    while total < n:
        run = int(rng.integers(1, 5000)) if rng.random() < 0.02 else int(rng.integers(1, 6))
        parts.append(chr(97 + int(rng.integers(3))) * run); total += run
    text = "".join(parts)[:n]
    want = O.byte_pair_encoding(text, 60, fast=True)
    try:
        trainer.set_train_grid(grid)
        got = _train(text, 60)
    finally:
        trainer.set_train_grid(0)
    assert list(got[0]) == list(want[0]) and got[2] == want[2]
