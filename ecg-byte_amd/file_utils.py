"""Mirror of ecg_byte/utils/file_utils.py (same names, arguments and return values): the on-disk layout the
reference's preprocessing leaves behind and main.py consumes --
  data/<dataset>/ecg/<split>/ecg_{i}_{j}.npy    float64 (12, seg_len) segment j of record i   (preprocess_utils.py:215-226)
  data/<dataset>/text/<split>/text_{i}_{j}.json the paired report / conversation / QA triple  (preprocess_utils.py:773-792)
  data/<tokenizer>.pkl                          pickle((vocab, merges))                       (tokenizer_utils.py:62-69)
  percentiles .npy                              pickled dict {'percentile_1', 'percentile_99'} (preprocess_utils.py:208-210)
Host-side only (paths, json, pickle): nothing here touches the GPU."""
from __future__ import annotations

import glob
import json
import os
import pickle
import random
import re

import numpy as np


def ensure_directory_exists(directory_path):
    """file_utils.py:10-15"""
    try:
        os.makedirs(directory_path, exist_ok=True)
        print(f"Directory ensured: {directory_path}")
    except Exception as e:
        print(f"Error ensuring directory {directory_path}: {str(e)}")


def load_vocab_and_merges(filename):
    """file_utils.py:17-20"""
    with open(filename, "rb") as f:
        vocab, merges = pickle.load(f)
    return vocab, merges


def open_json(path_to_file):
    """file_utils.py:22-24"""
    with open(path_to_file) as json_file:
        return json.load(json_file)


def load_npy(file_path):
    """file_utils.py:26-27"""
    return np.load(file_path)


def align_signal_text_files(signal_dir, text_dir):
    """Pairs `*.npy` and `*.json` files by the first `(\\d+)_(\\d+)` in their base names and returns the two path lists
    sorted by that (record, segment) index (file_utils.py:30-48).  Files without a partner are dropped."""
    signal_files = glob.glob(os.path.join(signal_dir, "*.npy"))
    text_files = glob.glob(os.path.join(text_dir, "*.json"))

    def extract_indices(filename):
        match = re.search(r"(\d+)_(\d+)", os.path.basename(filename))
        return tuple(map(int, match.groups())) if match else None

    signal_dict = {extract_indices(f): f for f in signal_files if extract_indices(f)}
    text_dict = {extract_indices(f): f for f in text_files if extract_indices(f)}
    common = sorted(set(signal_dict) & set(text_dict))
    return [signal_dict[i] for i in common], [text_dict[i] for i in common]


def sample_N_percent_indices(length, N=0.1):
    """file_utils.py:51-53 (uses the `random` module's global state, as the reference does)."""
    sample_size = max(1, int(length * N))
    return random.sample(range(length), sample_size)


def sample_N_percent_from_lists(list1, list2=None, N=0.05):
    """file_utils.py:55-64"""
    if list2 is not None and len(list1) != len(list2):
        raise ValueError("Both lists must have the same length")
    idx = sample_N_percent_indices(len(list1), N)
    s1 = [list1[i] for i in idx]
    if list2 is None:
        return s1
    return s1, [list2[i] for i in idx]


def save_percentiles(path, percentile_1, percentile_99):
    """The writer side of `np.load(args.percentiles, allow_pickle=True).item()` (data_loader.py:47):
    preprocess_utils.py:208-210 saves a dict with np.save."""
    np.save(path, {"percentile_1": float(percentile_1), "percentile_99": float(percentile_99)})


def load_percentiles(path):
    return np.load(path, allow_pickle=True).item()
