"""Mirror of the reference CLI ecg_byte/train_tokenizer.py (same flags, 8-16; same outputs, 19-66)
with corpus building, BPE training and the self-check encode on the MI355X.

    python -m ecg_byte_amd.train_tokenizer --train --num_merges 3500 --sampled_files F --percentiles P
"""
from __future__ import annotations

import argparse
import time

import numpy as np

from . import rust_bpe
from .tokenizer_utils import (decode_text, encode_text, load_vocab_and_merges, process_ecg,
                              process_large_file, reverse_normalize_all, save_vocab_and_merges)


def get_args(argv=None):
    parser = argparse.ArgumentParser(description=None)
    parser.add_argument("--num_merges", type=int, default=3500, help="Please choose the vocabulary size")
    parser.add_argument("--sampled_files", type=str, default=None, help="Path to the .txt file of sampled ecgs")
    parser.add_argument("--num_processes", type=int, default=2, help="Accepted for compatibility (GPU path ignores it)")
    parser.add_argument("--percentiles", type=str, default=None, help="Path to the calculated percentiles")
    parser.add_argument("--train", action="store_true", default=None, help="Train the tokenizer")
    parser.add_argument("--loaded", type=str, default=None, help="Path to an existing .pkl tokenizer")
    parser.add_argument("--check_ecg", type=str, default="./data/seg_ecg_qa_ptb_500/ecg/train/ecg_10_1.npy",
                        help="Record used for the encode/decode self-check (the reference hard-codes this path)")
    parser.add_argument("--out", type=str, default=None, help="Output .pkl (default ./data/tokenizer_{num_merges}.pkl)")
    return parser.parse_args(argv)


def main(args):
    percentiles = np.load(args.percentiles, allow_pickle=True).item()
    tokenizer_file_name = args.out or f"./data/tokenizer_{args.num_merges}.pkl"
    if args.train:
        all_string_signals = process_large_file(args.sampled_files, percentiles, args.num_processes)
        print(f"Total ECGs processed: {len(all_string_signals)}")
        print(list(all_string_signals)[:100])
        start_time = time.time()
        ids, vocab, merges = rust_bpe.byte_pair_encoding(all_string_signals, args.num_merges, args.num_processes)
        print(f"Byte pair encoding executed in {time.time() - start_time:.2f} seconds")
        print("Shared vocabulary across all ECGs:")
        print(f"Original length: {len(all_string_signals)}")
        print(f"Encoded length: {len(ids)}")
        print(f"Compression ratio: {len(all_string_signals) / max(1, len(ids)):.2f}X")
        print(f"Vocabulary size: {len(vocab)}")
        save_vocab_and_merges(vocab, merges, tokenizer_file_name)
        print(f"Vocabulary and merges saved to {tokenizer_file_name}")
    if args.loaded is None:
        args.loaded = tokenizer_file_name
    loaded_vocab, loaded_merges = load_vocab_and_merges(args.loaded)
    print(f"Loaded vocabulary and merges from {args.loaded}")

    new_ecg_signal = np.load(args.check_ecg)
    new_ecg_text = process_ecg(args.check_ecg, percentiles=percentiles)
    print(f"Processed ECG signal to text (first 100 characters): {new_ecg_text[:100]}...")
    print(f"Total tokens: {len(new_ecg_text)}")
    encoded_ecg = encode_text(new_ecg_text, loaded_merges)
    print(f"Encoded ECG (first 20 tokens): {encoded_ecg[:20]}...")
    print(f"Total tokens: {len(encoded_ecg)}")
    print(f"Compression ratio: {len(new_ecg_text) / max(1, len(encoded_ecg)):.2f}X")
    decoded_text = decode_text(encoded_ecg, loaded_vocab)
    print(f"Decoded text (first 100 characters): {decoded_text[:100]}...")
    print(decoded_text == new_ecg_text)
    decoded_signal = reverse_normalize_all(np.array(list(decoded_text)).reshape(new_ecg_signal.shape), percentiles)
    print(f"Maximum difference between original and decoded: {np.max(np.abs(new_ecg_signal - decoded_signal))}")
    return decoded_text == new_ecg_text


if __name__ == "__main__":
    main(get_args())
