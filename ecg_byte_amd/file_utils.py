"""Host-side helpers for the on-disk layout the reference's preprocessing leaves behind and its `main.py` consumes (the
public names, arguments and return values of `ecg_byte/utils/file_utils.py`, so callers switch by changing the import):
  data/<dataset>/ecg/<split>/ecg_{i}_{j}.npy    float64 (12, seg_len) segment j of record i   (preprocess_utils.py:215-226)
  data/<dataset>/text/<split>/text_{i}_{j}.json the paired report / conversation / QA triple  (preprocess_utils.py:773-792)
  data/<tokenizer>.pkl                          pickle((vocab, merges))                       (tokenizer_utils.py:62-69)
  percentiles .npy                              pickled dict {'percentile_1', 'percentile_99'} (preprocess_utils.py:208-210)
Paths, json and pickle only: nothing here touches the GPU."""
from __future__ import annotations

import json
import pickle
import random
import re
from pathlib import Path

import numpy as np

_RECORD_SEGMENT = re.compile(r"(\d+)_(\d+)")      # ecg_{record}_{segment}.npy / text_{record}_{segment}.json


def ensure_directory_exists(directory_path):
    """Creates the directory (and parents); reports instead of raising, like the reference (file_utils.py:10-15)."""
    try:
        Path(directory_path).mkdir(parents=True, exist_ok=True)
    except OSError as err:
        print(f"could not create {directory_path}: {err}")
    else:
        print(f"directory ready: {directory_path}")


def load_vocab_and_merges(filename):
    """pickle((vocab, merges)) as `tokenizer_utils.save_vocab_and_merges` writes it (file_utils.py:17-20)."""
    vocab, merges = pickle.loads(Path(filename).read_bytes())
    return vocab, merges


def open_json(path_to_file):
    """file_utils.py:22-24"""
    return json.loads(Path(path_to_file).read_text())


def load_npy(file_path):
    """file_utils.py:26-27"""
    return np.load(file_path)


def _by_record_segment(directory, pattern):
    """{(record, segment): path} for the files of `directory` whose base name carries a `<digits>_<digits>` pair."""
    found = {}
    for path in Path(directory).glob(pattern):
        m = _RECORD_SEGMENT.search(path.name)
        if m:
            found[(int(m.group(1)), int(m.group(2)))] = str(path)
    return found


def align_signal_text_files(signal_dir, text_dir):
    """The `.npy` signals and `.json` texts that share a (record, segment) index, as two parallel path lists ordered
    by that index; files without a partner are left out (file_utils.py:30-48)."""
    signals = _by_record_segment(signal_dir, "*.npy")
    texts = _by_record_segment(text_dir, "*.json")
    keys = sorted(signals.keys() & texts.keys())
    return [signals[k] for k in keys], [texts[k] for k in keys]


def sample_N_percent_indices(length, N=0.1):
    """max(1, int(length * N)) distinct indices, drawn with the `random` module's global state as the reference does
    (file_utils.py:51-53): seeding `random` reproduces the reference's subset."""
    return random.sample(range(length), max(1, int(length * N)))


def sample_N_percent_from_lists(list1, list2=None, N=0.05):
    """The same random subset of one list or of two parallel lists (file_utils.py:55-64)."""
    if list2 is not None and len(list2) != len(list1):
        raise ValueError("Both lists must have the same length")
    picked = sample_N_percent_indices(len(list1), N)
    first = [list1[i] for i in picked]
    return first if list2 is None else (first, [list2[i] for i in picked])


def save_percentiles(path, percentile_1, percentile_99):
    """Writer side of `np.load(args.percentiles, allow_pickle=True).item()` (data_loader.py:47): the reference's
    preprocessing stores the dict with np.save (preprocess_utils.py:208-210)."""
    np.save(path, {"percentile_1": float(percentile_1), "percentile_99": float(percentile_99)})


def load_percentiles(path):
    return np.load(path, allow_pickle=True).item()
