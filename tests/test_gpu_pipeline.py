"""GPU tests of the path around the kernels, on the reference's on-disk layout: files ->
DeviceBatchLoader (threaded reads, pinned staging, one quantise+encode+assemble per batch) -> LLM wrapper ->
trainer / validater / tester runners -> checkpoint files.  Token-level results are compared with the oracle pipeline."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest

from helpers import WordTokenizer, load_tokenizer
from oracle import assemble as OA
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REPORTS = ["sinus rhythm normal ecg", "atrial fibrillation with rapid ventricular response", "sinus bradycardia otherwise normal ecg",
           "left bundle branch block", "sinus rhythm with premature ventricular complexes", "normal ecg"]


@pytest.fixture()
def ptb_dir(tmp_path):
    """data/<dataset>/{ecg,text}/<split>/ in the reference's naming (preprocess_utils.py:215-226, file_utils.py:30-48)."""
    from ecg_byte_amd import file_utils as F
    from ecg_byte_amd import synth
    vocab, merges, pc = load_tokenizer("c1")
    x = synth.synth_ecg(11, 1000, seed=5)
    for split, rng_ in (("train", range(0, 8)), ("val", range(8, 10)), ("test", range(10, 11))):
        os.makedirs(tmp_path / "ecg" / split); os.makedirs(tmp_path / "text" / split)
        for i in rng_:
            np.save(tmp_path / "ecg" / split / f"ecg_{i}_0.npy", x[i])
            (tmp_path / "text" / split / f"text_{i}_0.json").write_text(json.dumps(REPORTS[i % len(REPORTS)]))
    F.save_percentiles(tmp_path / "percentiles.npy", pc["percentile_1"], pc["percentile_99"])
    with open(tmp_path / "tok.pkl", "wb") as f:
        pickle.dump((vocab, merges), f)
    return tmp_path, x, pc


def _tokenizer(vocab):
    tok = WordTokenizer("Could you please help me explain my ECG?".split() + " ".join(REPORTS).split())
    tok.add_tokens([f"signal_{k}" for k in vocab.keys()])             # main.py:144-150
    tok.add_tokens(["<sig_start>"], special_tokens=True)
    tok.add_tokens(["<sig_end>"], special_tokens=True)
    tok.add_special_tokens({"pad_token": "<pad>"})
    return tok


def _dataset(root, split, inference=False, pad_to_max=124):
    from ecg_byte_amd import file_utils as F
    from ecg_byte_amd.data_loader import ECGTokenDataset
    vocab, merges = F.load_vocab_and_merges(str(root / "tok.pkl"))
    tok = _tokenizer(vocab)
    sig, txt = F.align_signal_text_files(str(root / "ecg" / split), str(root / "text" / split))
    args = SimpleNamespace(percentiles=str(root / "percentiles.npy"), dataset="ptb_500", inference=inference,
                           pad_to_max=pad_to_max, dis=False, dev=False, toy=True, device="cuda")
    return ECGTokenDataset(sig, txt, vocab, merges, tokenizer=tok, args=args), tok, args


def test_loader_batches_equal_per_sample_items_and_the_oracle(ptb_dir):
    from ecg_byte_amd.data_loader import DeviceBatchLoader
    root, x, pc = ptb_dir
    ds, tok, args = _dataset(root, "train")
    os.remove(ds.signal_path_list[5])                                   # an unreadable sample is dropped from its batch
    batches = list(DeviceBatchLoader(ds, batch_size=3, workers=3))
    assert [b["tokenized_signal"].shape[0] for b in batches] == [3, 2, 2] and len(DeviceBatchLoader(ds, batch_size=3)) == 3
    keep = [i for i in range(8) if i != 5]
    rows = {k: torch.cat([b[k] for b in batches]).cpu().numpy() for k in batches[0]}
    trie = O.Trie(ds.merges)
    keys = list(ds.vocab.keys())
    lut = {k: tok.convert_tokens_to_ids(f"signal_{k}") for k in keys}
    for n, i in enumerate(keep):
        item = ds[i]
        for k in rows:
            assert np.array_equal(rows[k][n], item[k].numpy()), (i, k)
        sig = [lut[int(t)] for t in trie.quantize_encode(x[i], pc["percentile_1"], pc["percentile_99"])]
        q = tok(["Could you please help me explain my ECG?"]).input_ids[0].tolist()
        a = tok([REPORTS[i % len(REPORTS)]]).input_ids[0].tolist()
        want = OA.prepare_training(sig, q, a, tok.pad_token_id, tok.convert_tokens_to_ids("<bos>"), tok.eos_token_id,
                                   tok.convert_tokens_to_ids("<sig_start>"), tok.convert_tokens_to_ids("<sig_end>"), args.pad_to_max)
        for k in want:
            assert np.array_equal(rows[k][n], np.asarray(want[k])), (i, k)
    assert ds[5] is None


def _tiny_model(tok, seed=0):
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    cfg = DecoderConfig(vocab_size=len(tok), hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=None, pad_token_id=tok.pad_token_id)
    return HipCausalLM(cfg, seed=seed)


def test_runners_train_validate_checkpoint_and_generate(ptb_dir, tmp_path):
    from ecg_byte_amd.data_loader import DeviceBatchLoader
    from ecg_byte_amd.llm import LLM
    from ecg_byte_amd.runners import tester, trainer, validater
    root, x, pc = ptb_dir
    train_ds, tok, args = _dataset(root, "train")
    val_ds, _, _ = _dataset(root, "val")
    model = LLM(_tiny_model(tok), args)
    opt = model.llm.make_optimizer(lr=1e-4, warmup=24)      # peak rate 128^-0.5 * 24^-0.5 = 0.018 at step 24, past the 12 steps of this run: no bounce
    run_dir = tmp_path / "run"; os.makedirs(run_dir)
    losses, vals = [], []
    for epoch in range(6):
        losses.append(trainer(model, DeviceBatchLoader(train_ds, batch_size=4, shuffle=True, seed=1), opt, args, epoch, str(run_dir),
                              checkpoint_every=2 if epoch == 0 else 50000)["average_loss"])
        vals.append(validater(model, DeviceBatchLoader(val_ds, batch_size=2), args, epoch)["average_loss"])

    assert min(losses) < 0.6 * losses[0] and min(vals) < 0.7 * vals[0] and vals[-1] < vals[0], (losses, vals)
    assert not os.path.exists(run_dir / "best_train_model_0_1.pth")        # args.toy suppresses the step checkpoints (train.py:34)
    # checkpoint format of main.py:299-306 and its reload (main.py:193-195)
    torch.save({"model": model.state_dict(), "epoch": 5}, run_dir / "best_model.pth")
    ck = torch.load(run_dir / "best_model.pth", map_location="cuda")
    assert ck["epoch"] == 5 and all(k.startswith("llm.") for k in ck["model"]) and "llm.model.embed_tokens.weight" in ck["model"]
    fresh = LLM(_tiny_model(tok, seed=9), args)
    fresh.load_state_dict(ck["model"])
    v2 = validater(fresh, DeviceBatchLoader(val_ds, batch_size=2), args, 0)["average_loss"]
    assert abs(v2 - vals[-1]) < 1e-5 * max(1.0, abs(v2))     # (the loss is summed with fp32 atomics)
    # the validation loss equals the training-path loss of the same batch (forward-only path vs autograd path)
    b = next(iter(DeviceBatchLoader(val_ds, batch_size=2)))
    with torch.no_grad():
        l_eval = fresh(b).loss.item()
    assert abs(fresh(b).loss.item() - l_eval) < 1e-5 * max(1.0, l_eval)
    # inference: batch-1 prompts without padding, greedy generate, metrics averaged over the samples
    test_ds, _, targs = _dataset(root, "test", inference=True)
    out = tester(fresh, DeviceBatchLoader(test_ds, batch_size=1), tok, targs)
    assert set(out["metrics"]) == {"BLEU"} and 0.0 <= out["metrics"]["BLEU"] <= 1.0
    assert out["qa_results"]["questions"] == ["Could you please help me explain my ECG?"] and len(out["qa_results"]["gen_answers"]) == 1
    p = next(iter(DeviceBatchLoader(test_ds, batch_size=1)))
    item = test_ds[0]
    assert torch.equal(p["tokenized_signal"][0].cpu(), item["tokenized_signal"]) and p["attn_mask"].shape == p["tokenized_signal"].shape


def _write_model_dir(path, vocab_words):
    """A local checkpoint directory in the hub layout: tiny Llama config + safetensors + a word-level fast tokenizer."""
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    words = ["<unk>", "<bos>", "<eos>"] + sorted(set(vocab_words))
    t = Tokenizer(WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    os.makedirs(path, exist_ok=True)
    t.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<bos>", "eos_token": "<eos>", "unk_token": "<unk>"}, f)
    cfg = DecoderConfig(vocab_size=len(words), hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=None)
    m = HipCausalLM(cfg, seed=4)
    m.save_pretrained(path)
    return m


def test_pretrained_directory_round_trip(tmp_path):
    from ecg_byte_amd.decoder import HipCausalLM
    m = _write_model_dir(str(tmp_path / "tiny"), ["a", "b"])
    assert sorted(os.listdir(tmp_path / "tiny")) == ["config.json", "model.safetensors", "tokenizer.json", "tokenizer_config.json"]
    cfg = json.load(open(tmp_path / "tiny" / "config.json"))
    assert cfg["model_type"] == "llama" and cfg["num_key_value_heads"] == 1 and cfg["tie_word_embeddings"] is True
    m2 = HipCausalLM.from_pretrained(str(tmp_path / "tiny"))
    a, b = m.state_dict(), m2.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_main_cli_trains_then_infers_on_the_reference_layout(ptb_dir, tmp_path, monkeypatch):
    """ecg_byte_amd.main with the reference's options on a data/ + runs/ tree: two epochs (--dev), best_model.pth in the
    reference's run directory, then --inference over the five seeds with the per-seed and statistics files."""
    pytest.importorskip("transformers")
    from ecg_byte_amd import main as M
    root, x, pc = ptb_dir
    data = tmp_path / "data"
    os.makedirs(data / "ptb_500")
    os.rename(root / "ecg", data / "ptb_500" / "ecg")
    os.rename(root / "text", data / "ptb_500" / "text")
    os.rename(root / "tok.pkl", data / "tokenizer_c1.pkl")
    model_dir = str(tmp_path / "tiny-llama")
    _write_model_dir(model_dir, "Could you please help me explain my ECG ?".split() + " ".join(REPORTS).split())
    common = ["--device", "cuda:0", "--model", model_dir, "--dataset", "ptb_500", "--tokenizer_check", "tokenizer_c1",
              "--percentiles", str(root / "percentiles.npy"), "--pad_to_max", "124", "--batch_size", "4", "--warmup", "4",
              "--data_root", str(data), "--runs_root", str(tmp_path / "runs"), "--num_merges", "1000", "--peft"]
    out = M.main(common + ["--dev"])
    assert len(out["train_loss"]) == 2 and all(np.isfinite(out["train_loss"])) and out["val_loss"][1] < out["val_loss"][0]
    assert os.path.exists(os.path.join(out["directory"], "best_model.pth"))
    ck = torch.load(os.path.join(out["directory"], "best_model.pth"), map_location="cpu")
    assert ck["epoch"] in (0, 1) and any("lora_A" in k for k in ck["model"])          # --peft: adapters are in the checkpoint
    rel = os.path.relpath(out["directory"], tmp_path / "runs" / "0")
    stats = M.main(common + ["--inference", "--checkpoint", rel])
    assert set(stats) == {"BLEU"} and len(stats["BLEU"]["raw_values"]) == 5
    files = os.listdir(out["directory"])
    assert "statistical_analysis_ptb_500.json" in files and sum(f.startswith("seed_") for f in files) == 5


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (rank-sharded records, barrier + max-over-ranks timing, gradient all-reduce in the train
    leg) end to end: two ranks on cuda:0 over gloo, through bench.py's test hooks -- a one-GPU box cannot run RCCL
    between two ranks, the driver's 8-GPU run can."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ECGB_BENCH_BACKEND="gloo", ECGB_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "512", "--steps", "3",
           "--warmup", "1", "--train-batch", "2", "--train-steps", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["config"]["records_per_gpu"] == 512
    # both ranks' tokens are in the aggregate: twice one rank's records
    assert abs(out["records_per_s"] * out["ms_per_step"] * 1e-3 - 2 * 512) < 1e-4 * 1024
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0
    assert out["train"]["steps"] == 2 and np.isfinite(out["train"]["final_loss"]) and "dp2" in out["train"]["config"]["parallelism"]
    assert np.isfinite(out["train"]["lora_r16"]["final_loss"])


def test_bench_line_contract_single_gpu():
    """The one JSON line bench.py prints (N = 1, encode leg with its CPU baseline; the train leg is skipped here, the
    two-rank test above runs it): every key the driver and SURVEY 8d ask for, with consistent values."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-train"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["metric"] == "ecg_tokens_per_sec_encode" and out["unit"] == "tokens/s" and out["n_gpus"] == 1
    assert out["steps"] == 5 and out["warmup"] == 2 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic" and "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"] and 0 < r["frac"] < 1      # (the line's floats carry six significant digits)
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.9
    # whole-job value = tokens of one launch / time of one launch
    n_tok = out["tokens_per_record"] * out["config"]["records_per_gpu"]
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - n_tok) < 1e-4 * n_tok
    assert len(lines[0]) < 8192, "the driver keeps 8 KB of the line"
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "tokens/s" and c["value"] > 0 and "records" in c["sample"]
    assert c["quantiser_python_style_symbols_per_s"] < c["quantiser_c_symbols_per_s"]
