"""Tokenizer training CLI: the flags of the reference's `ecg_byte/train_tokenizer.py` (8-16), on the MI355X path.

    python -m ecg_byte_amd.train_tokenizer --train --num_merges 3500 --sampled_files F --percentiles P

Two stages, either of which can run alone:
  train       corpus string from the sampled .npy records (quantise on the device) -> `rust_bpe.byte_pair_encoding`
              (HIP trainer) -> pickle((vocab, merges))                                     [reference: 19-42]
  self-check  one record: quantise, encode with the stored merges, decode, compare         [reference: 44-66]
`main` returns True when the decoded text equals the quantised text -- the reference prints that comparison; the tests
use it.
"""
from __future__ import annotations

import argparse
import time
from dataclasses import dataclass

import numpy as np

from . import rust_bpe
from . import tokenizer_utils as tu

# (flag, type / action, default, help) -- the reference's options, plus --check_ecg / --out for what it hard-codes
_OPTIONS = (
    ("--num_merges", int, 3500, "number of BPE merges (vocabulary size - 256)"),
    ("--sampled_files", str, None, "text file listing the sampled .npy records, one path per line"),
    ("--num_processes", int, 2, "accepted for compatibility; the device path does not fork workers"),
    ("--percentiles", str, None, "pickled-dict .npy with percentile_1 / percentile_99"),
    ("--train", "store_true", None, "train a tokenizer before the self-check"),
    ("--loaded", str, None, "existing tokenizer .pkl to check instead of the one just trained"),
    ("--check_ecg", str, "./data/seg_ecg_qa_ptb_500/ecg/train/ecg_10_1.npy", "record used by the self-check"),
    ("--out", str, None, "where to write the tokenizer (default ./data/tokenizer_{num_merges}.pkl)"),
    ("--dis", "store_true", None, "started by torch.distributed.run, one process per GPU: every rank quantises its contiguous share of "
                                  "the listed records and trains on that slice of the corpus (RCCL; same merges as one GPU)"),
)


def get_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, kind, default, text in _OPTIONS:
        if kind == "store_true":
            ap.add_argument(flag, action="store_true", default=default, help=text)
        else:
            ap.add_argument(flag, type=kind, default=default, help=text)
    return ap.parse_args(argv)


@dataclass
class TrainReport:
    symbols: int
    tokens: int
    vocab_size: int
    seconds: float

    def lines(self):
        ratio = self.symbols / max(1, self.tokens)
        yield f"corpus: {self.symbols} symbols -> {self.tokens} tokens ({ratio:.2f}x), vocabulary {self.vocab_size}"
        yield f"byte_pair_encoding: {self.seconds:.2f} s"


def train(sampled_files, percentiles, num_merges, num_processes, out_path) -> TrainReport:
    corpus = tu.process_large_file(sampled_files, percentiles, num_processes)
    t0 = time.time()
    ids, vocab, merges = rust_bpe.byte_pair_encoding(corpus, num_merges, num_processes)
    dt = time.time() - t0
    tu.save_vocab_and_merges(vocab, merges, out_path)
    return TrainReport(symbols=len(corpus), tokens=len(ids), vocab_size=len(vocab), seconds=dt)


def train_sharded(sampled_files, percentiles, num_merges, out_path) -> TrainReport:
    """The same on a corpus split over the ranks of the default process group: the listed records are
    dealt out in contiguous runs (the corpus is their concatenation IN FILE ORDER, tokenizer_utils.py:79-93), each rank quantises
    its run and keeps the ids of its slice; rank 0 writes the pickle."""
    import torch
    import torch.distributed as dist
    from .trainer import HipShard, bpe_train_sharded
    rank, world = dist.get_rank(), dist.get_world_size()
    with open(sampled_files) as f:
        paths = [line.strip() for line in f if line.strip()]
    lo, hi = len(paths) * rank // world, len(paths) * (rank + 1) // world
    part = tu.process_paths(paths[lo:hi], percentiles)
    text = torch.frombuffer(bytearray(part), dtype=torch.uint8).cuda() if part else torch.empty(0, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    ids, pairs = bpe_train_sharded(HipShard(text, num_merges), num_merges)
    dt = time.time() - t0
    counts = torch.tensor([len(part), len(ids)], dtype=torch.int64, device="cuda")
    dist.all_reduce(counts)
    vocab, merges = rust_bpe.vocab_merges_from_pairs(pairs)
    if rank == 0:
        tu.save_vocab_and_merges(vocab, merges, out_path)
    dist.barrier()
    return TrainReport(symbols=int(counts[0]), tokens=int(counts[1]), vocab_size=len(vocab), seconds=dt)


def self_check(tokenizer_path, record_path, percentiles) -> bool:
    """encode -> decode of one record must give its quantised text back (train_tokenizer.py:58-60)."""
    vocab, merges = tu.load_vocab_and_merges(tokenizer_path)
    signal = np.load(record_path)
    text = tu.process_ecg(record_path, percentiles=percentiles)
    ids = tu.encode_text(text, merges)
    back = tu.decode_text(ids, vocab)
    same = back == text
    print(f"self-check on {record_path}: {len(text)} symbols -> {len(ids)} tokens ({len(text) / max(1, len(ids)):.2f}x); "
          f"first ids {list(ids[:12])}; round trip {'ok' if same else 'MISMATCH'}")
    if same:
        rebuilt = tu.reverse_normalize_all(np.array(list(back)).reshape(signal.shape), percentiles)
        print(f"largest |signal - dequantised| = {float(np.max(np.abs(signal - rebuilt))):.6g}")
    return same


def main(args) -> bool:
    percentiles = np.load(args.percentiles, allow_pickle=True).item()
    target = args.out or f"./data/tokenizer_{args.num_merges}.pkl"
    rank = 0
    if args.dis:
        import os
        import torch
        import torch.distributed as dist
        rank = int(os.environ["RANK"])
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group("nccl")                                    # RCCL
    try:
        if args.train:
            report = (train_sharded(args.sampled_files, percentiles, args.num_merges, target) if args.dis else
                      train(args.sampled_files, percentiles, args.num_merges, args.num_processes, target))
            if rank == 0:
                for line in report.lines():
                    print(line)
                print(f"tokenizer written to {target}")
        return self_check(args.loaded or target, args.check_ecg, percentiles) if rank == 0 else True
    finally:
        if args.dis:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    raise SystemExit(0 if main(get_args()) else 1)
