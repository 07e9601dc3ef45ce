"""CPU tests of the host-side mirrors around the hot path: on-disk layout helpers (file_utils), BLEU / statistics
(model_utils) and the runners' bookkeeping.  BLEU has no reference build to compare with (nltk is not installed):
the expected values below are computed by hand from NLTK's published definition."""
import json
import math
import os
import random

import numpy as np
import pytest


def test_align_signal_text_files_pairs_by_record_and_segment(tmp_path):
    from ecg_byte_amd.file_utils import align_signal_text_files
    sd, td = tmp_path / "ecg", tmp_path / "text"
    sd.mkdir(); td.mkdir()
    for i, j in [(10, 0), (2, 1), (2, 0), (7, 3)]:
        np.save(sd / f"ecg_{i}_{j}.npy", np.zeros((12, 4)))
    for i, j in [(2, 0), (10, 0), (2, 1), (99, 0)]:
        (td / f"text_{i}_{j}.json").write_text(json.dumps("x"))
    (td / "notes.json").write_text("{}")                       # no indices: ignored
    s, t = align_signal_text_files(str(sd), str(td))
    assert [os.path.basename(p) for p in s] == ["ecg_2_0.npy", "ecg_2_1.npy", "ecg_10_0.npy"]   # numeric, not lexicographic
    assert [os.path.basename(p) for p in t] == ["text_2_0.json", "text_2_1.json", "text_10_0.json"]


def test_sampling_follows_the_random_module(tmp_path):
    from ecg_byte_amd import file_utils as F
    a, b = list(range(100)), [str(i) for i in range(100)]
    random.seed(3)
    want = random.sample(range(100), 25)
    random.seed(3)
    sa, sb = F.sample_N_percent_from_lists(a, b, 0.25)
    assert sa == want and sb == [str(i) for i in want]
    assert len(F.sample_N_percent_from_lists([1, 2, 3], N=0.01)) == 1          # at least one
    with pytest.raises(ValueError):
        F.sample_N_percent_from_lists([1, 2], [1], 0.5)


def test_percentiles_and_tokenizer_files_round_trip(tmp_path):
    from ecg_byte_amd import file_utils as F
    from ecg_byte_amd.tokenizer_utils import save_vocab_and_merges
    F.save_percentiles(tmp_path / "p.npy", -0.5, 1.25)
    assert np.load(tmp_path / "p.npy", allow_pickle=True).item() == {"percentile_1": -0.5, "percentile_99": 1.25}   # data_loader.py:47
    vocab, merges = {97: "a", 256: "ab"}, [([97, 98], 256)]
    save_vocab_and_merges(vocab, merges, str(tmp_path / "tok.pkl"))
    assert F.load_vocab_and_merges(str(tmp_path / "tok.pkl")) == (vocab, merges)


def test_bleu_known_answers():
    from ecg_byte_amd.model_utils import calculate_bleu, corpus_bleu
    assert calculate_bleu(["the cat sat on the mat"], ["the cat sat on the mat"]) == pytest.approx(1.0)
    # p1 = 5/6, p2 = 3/5, p3 = 1/4, p4 = 0 -> method1: 0.1/3; equal lengths -> BP = 1
    want = math.exp(0.25 * (math.log(5 / 6) + math.log(3 / 5) + math.log(1 / 4) + math.log(0.1 / 3)))
    assert calculate_bleu(["the cat sat on the mat"], ["the cat is on the mat"]) == pytest.approx(want, rel=1e-12)
    assert calculate_bleu(["a b c"], ["x y z"]) == 0                       # no unigram match: 0, not smoothed
    # brevity penalty: hypothesis 2 tokens, reference 4; p1 = 2/2, p2 = 1/1, p3, p4: no n-grams -> 0/max(1,0) -> eps/1
    want = math.exp(1 - 4 / 2) * math.exp(0.25 * (0 + 0 + 2 * math.log(0.1)))
    assert calculate_bleu(["a b c d"], ["a b"]) == pytest.approx(want, rel=1e-12)
    # corpus level: counts are pooled over sentences before the ratio
    refs, hyps = [["a b c d".split()], ["e f g h".split()]], ["a b c d".split(), "e f x h".split()]
    p = [7 / 8, 4 / 6, 2 / 4, 1 / 2]
    assert corpus_bleu(refs, hyps) == pytest.approx(math.exp(sum(0.25 * math.log(x) for x in p)), rel=1e-12)
    # clipping: a repeated word counts at most as often as in the reference
    assert calculate_bleu(["the cat"], ["the the"]) == pytest.approx(math.exp(0.25 * (math.log(1 / 2) + math.log(0.1 / 1) + 2 * math.log(0.1))), rel=1e-12)


def test_statistics_and_early_stopping():
    from ecg_byte_amd.model_utils import early_stopping, run_statistical_analysis
    from scipy import stats
    res = [{"metrics": {"BLEU": v, "hf-f1": 2 * v}} for v in (0.10, 0.12, 0.11, 0.15, 0.09)]
    out = run_statistical_analysis(res)
    vals = np.array([10, 12, 11, 15, 9], dtype=float)
    assert out["BLEU"]["mean"] == pytest.approx(vals.mean()) and out["BLEU"]["std"] == pytest.approx(vals.std(ddof=1))
    half = stats.t.ppf(0.975, 4) * vals.std(ddof=1) / math.sqrt(5)
    assert out["BLEU"]["conf_interval"] == pytest.approx((vals.mean() - half, vals.mean() + half))
    assert out["hf-f1"]["raw_values"] == pytest.approx(list(2 * vals))
    assert not early_stopping([3, 2, 1], patience=5)
    assert early_stopping([1.0, 1.2, 1.3, 1.4, 1.5, 1.6], patience=5, delta=0.01)
    assert not early_stopping([1.0, 1.2, 1.3, 1.4, 1.5, 1.005], patience=5, delta=0.01)


def test_tester_aggregates_like_the_reference():
    """inference.py:52-66 with a stub model: per-sample BLEU averaged, sub-dict metrics flattened, failures count as 0."""
    from types import SimpleNamespace
    from ecg_byte_amd.runners import tester

    class Stub:
        def __init__(self): self.k = 0
        def eval(self): pass
        def generate(self, batch, tokenizer):
            self.k += 1
            if self.k == 3:
                raise RuntimeError("boom")
            return batch["answer"][0] if self.k == 1 else "nothing in common"
    data = [{"answer": [a], "question": [q]} for a, q in (("sinus rhythm normal ecg", "q1"), ("atrial fibrillation", "q2"), ("x", "q3"))]
    extra = lambda refs, hyps: {"ROUGE": {"rouge-1": 0.5, "rouge-l": 0.25}, "BERTSCORE": {"hf-f1": [0.75]}}
    out = tester(Stub(), data + [None], None, SimpleNamespace(dev=False, device="cpu"), extra_metrics=extra)
    assert out["metrics"]["BLEU"] == pytest.approx((1.0 + 0 + 0) / 3)
    # the failed sample scores zero in EVERY metric (inference.py:34-39), not only in BLEU
    assert out["metrics"]["rouge-1"] == pytest.approx(2 * 0.5 / 3) and out["metrics"]["hf-f1"] == pytest.approx(2 * 0.75 / 3)
    assert out["metrics"]["METEOR"] == 0 and out["metrics"]["rouge-2"] == 0
    assert out["qa_results"] == {"questions": ["q1", "q2"], "gt_answers": ["sinus rhythm normal ecg", "atrial fibrillation"],
                                 "gen_answers": ["sinus rhythm normal ecg", "nothing in common"]}


def test_global_stats_and_segment_files_are_what_the_reference_writes(tmp_path):
    """compute_global_stats + process_and_save_batch (the driver hop of preprocess_utils.py:168-226 around the conditioning): the statistics against a literal numpy
    restatement of the reference's loop under the same np.random seed, the files against its np.save / json.dump calls, and ECGTokenDataset's own readers
    (np.load of the signal, allow_pickle load of the percentiles dict) on what was written.  Runs on CPU tensors: the functions take any device."""
    import torch
    from ecg_byte_amd import preprocess_utils as pp
    from ecg_byte_amd.file_utils import align_signal_text_files, load_percentiles
    rng = np.random.default_rng(5)
    batches = [rng.standard_normal((3, 2, 1250, 12)), rng.standard_normal((4, 2, 1250, 12)) * 2.0, rng.standard_normal((5, 2, 1250, 12))]
    # the reference's loop, literally (preprocess_utils.py:189-207), over the same instances in the same order
    np.random.seed(11)
    gmin, gmax, samples, total = np.inf, -np.inf, [], 0
    for b in batches:
        for inst in b:
            for seg in inst:
                gmin, gmax = min(gmin, np.min(seg)), max(gmax, np.max(seg))
                if total < 100000:
                    k = min(100000 - total, seg.size)
                    samples.extend(seg.flat[np.random.choice(seg.size, k, replace=False)])
                    total += k
    want = {"global_min": gmin, "global_max": gmax, "percentile_1": np.percentile(np.array(samples), 1), "percentile_99": np.percentile(np.array(samples), 99),
            "skipped_instances": 2}
    np.random.seed(11)
    got = pp.compute_global_stats([torch.from_numpy(b) for b in batches], skipped=2)
    assert got == want and total == 100000
    pp.save_dataset_stats(tmp_path / "ptb_dataset_stats.npy", got)
    pc = load_percentiles(tmp_path / "ptb_dataset_stats.npy")                       # data_loader.py:47's np.load(...).item()
    assert pc["percentile_1"] == want["percentile_1"] and pc["percentile_99"] == want["percentile_99"]
    # files: instance 1 of 4 was dropped by the conditioning (raw NaN): indices 10, 12, 13 are written, 11 is not
    kept = torch.tensor([True, False, True, True])
    seg = torch.from_numpy(batches[0])
    texts = [["q-type", f"question {i}", ["answer", str(i)]] for i in range(4)]
    wrote = pp.process_and_save_batch(seg, texts, 10, "train", "ecg_qa_ptb", 1250, root=str(tmp_path / "data"), kept=kept)
    assert wrote == [10, 12, 13]
    sig, txt = align_signal_text_files(str(tmp_path / "data" / "ecg_qa_ptb_1250" / "ecg" / "train"), str(tmp_path / "data" / "ecg_qa_ptb_1250" / "text" / "train"))
    assert [os.path.basename(p) for p in sig] == [f"ecg_{i}_{j}.npy" for i in (10, 12, 13) for j in (0, 1)]
    a = np.load(sig[2])                                                             # ecg_12_0 = record 1 of the kept ones, segment 0
    assert a.shape == (12, 1250) and a.dtype == np.float64 and np.array_equal(a, batches[0][1, 0].T)
    assert np.isfortran(a)                                                          # the reference saves the transposed VIEW: the header says fortran_order
    assert json.load(open(txt[2])) == texts[2]
    with pytest.raises(ValueError):
        pp.process_and_save_batch(seg, texts, 0, "train", "x", 1250, root=str(tmp_path / "d2"), kept=torch.tensor([True, True, True, True]))


def test_lora_site_initialises_as_peft_does_kaiming_uniform_a_and_zero_b():
    """peft's LoraLayer.reset_lora_parameters (the reference's get_peft_model, main.py:131-155): lora_A by
    nn.init.kaiming_uniform_(a=sqrt(5)) -- torch's own rule gives U(-1/sqrt(in), 1/sqrt(in)) for an [r, in] weight -- lora_B zeros, so a fresh adapter
    changes nothing.  The stacked site must draw every live row from exactly that interval (and fill it), keep its padding rows zero, and start with B = 0."""
    import math
    import torch
    from ecg_byte_amd.decoder import LoraSite
    in_dim, n_sub, r = 512, 3, 16
    ref = torch.empty(r, in_dim)
    torch.nn.init.kaiming_uniform_(ref, a=math.sqrt(5))                 # what peft calls
    gain = torch.nn.init.calculate_gain("leaky_relu", math.sqrt(5))
    bound = gain * math.sqrt(3.0 / in_dim)
    assert abs(bound - 1.0 / math.sqrt(in_dim)) < 1e-12 and float(ref.abs().max()) <= bound
    site = LoraSite(in_dim, [(0, 64), (64, 32), (96, 32)], r, 32, 0.05, "cpu", torch.Generator().manual_seed(3))
    A, B = site.A.detach().float(), site.B.detach().float()
    assert A.shape == (64, in_dim) and B.shape == (128, 64) and site.scale == 2.0
    live = A[: r * n_sub]
    assert float(live.abs().max()) <= bound * (1 + 2 ** -8)             # (bf16 rounding of a value at the bound)
    assert float(live.abs().max()) > 0.98 * bound                       # the interval is filled ...
    assert abs(float(live.std()) - bound / math.sqrt(3.0)) < 0.02 * bound and abs(float(live.mean())) < 0.02 * bound    # ... uniformly
    assert float(A[r * n_sub:].abs().max()) == 0.0 and float(B.abs().max()) == 0.0
    # every block of B may only ever hold its own 16 columns
    m = site.bmask.float()
    assert m.sum().item() == 128 * r and all(float(m[lo:hi, 16 * b: 16 * b + 16].min()) == 1.0 for b, (lo, hi) in enumerate([(0, 64), (64, 96), (96, 128)]))
