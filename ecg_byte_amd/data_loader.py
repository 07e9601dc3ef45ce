"""Mirror of the reference's ECGTokenDataset (ecg_byte/data_loader.py:34-132) with the
per-sample Python work -- normalize_all, ''.join, rust_bpe.encode_text, the 'signal_{id}' ->
LLM-id mapping and _prepare_training -- done on the MI355X for a whole batch at once.

Two surfaces:
  * `ECGTokenDataset`   same constructor / __getitem__ dict as the reference (drop-in);
  * `BatchAssembler`    the batched fast path: `(B,12,L)` float64 signals on the device +
                        per-sample question/answer ids -> the four `[B, pad_to_max+4]` tensors
                        `LLM.forward` consumes (ecg_byte/models/llm.py:18-21), on the device.
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np
import torch

from . import _lib
from .tokenizer import HipTokenizer, _ptr, _stream_ptr


def _as_i32_concat(seqs, device):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.zeros(max(1, int(off[-1])), dtype=np.int32)
    for i, s in enumerate(seqs):
        flat[off[i]:off[i + 1]] = s
    return (torch.from_numpy(flat).to(device), torch.from_numpy(off.astype(np.int32)).to(device), lens)


class BatchAssembler:
    """Quantise -> encode -> map -> assemble for a batch, entirely on the device."""

    def __init__(self, merges, signal_token_lut, pad_id, bos_id, eos_id, sig_start_id, sig_end_id,
                 pad_to_max, device="cuda"):
        self.tok = merges if isinstance(merges, HipTokenizer) else HipTokenizer(merges)
        lut = np.asarray(signal_token_lut, dtype=np.int32)
        self.device = torch.device(device)
        self.lut = torch.from_numpy(lut).to(self.device)
        self.pad_id, self.bos_id, self.eos_id = int(pad_id), int(bos_id), int(eos_id)
        self.sig_start_id, self.sig_end_id = int(sig_start_id), int(sig_end_id)
        self.pad_to_max = int(pad_to_max)

    def encode(self, signal, percentiles, max_tokens=None):
        stride = int(max_tokens) if max_tokens else None
        return self.tok.quantize_encode(signal, percentiles, ids_stride=stride)

    def assemble(self, ids, counts, questions, answers=None, inference=False):
        """ids/counts: encoder output.  questions/answers: per-sample lists of LLM ids.
        Training -> dict of `[B, pad_to_max+4]` device tensors with the reference's keys
        (data_loader.py:126-131); inference -> `tokenized_signal`, `attn_mask`, `lengths`."""
        B = ids.shape[0]
        q_flat, q_off, q_len = _as_i32_concat(questions, self.device)
        if inference:
            a_flat = a_off = None
            row_len = int(2 + min(int(counts.max().item()), ids.shape[1]) + 1 + (q_len.max() if B else 0))
        else:
            a_flat, a_off, a_len = _as_i32_concat(answers, self.device)
            over = np.nonzero(q_len + a_len > self.pad_to_max)[0]
            if over.size:   # the reference dies on its length assert here (data_loader.py:123)
                raise AssertionError(
                    f"Lengths don't match: sample {int(over[0])} has len(question)+len(answer) = "
                    f"{int(q_len[over[0]] + a_len[over[0]])} > pad_to_max = {self.pad_to_max}")
            row_len = self.pad_to_max + 4
        dev = self.device
        input_ids = torch.empty((B, row_len), dtype=torch.int64, device=dev)
        attn = torch.empty((B, row_len), dtype=torch.float32, device=dev)
        labels = pos = lengths = None
        if inference:
            lengths = torch.empty((B,), dtype=torch.int32, device=dev)
        else:
            labels = torch.empty((B, row_len), dtype=torch.int64, device=dev)
            pos = torch.empty((B, row_len), dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().ecgb_assemble_hip(
            _ptr(ids), ids.shape[1], _ptr(counts), B, _ptr(self.lut), self.lut.numel(),
            _ptr(q_flat), _ptr(q_off), _ptr(a_flat), _ptr(a_off),
            self.pad_id, self.bos_id, self.eos_id, self.sig_start_id, self.sig_end_id,
            self.pad_to_max, 1 if inference else 0, row_len,
            _ptr(input_ids), _ptr(attn), _ptr(labels), _ptr(pos), _ptr(lengths), _stream_ptr()))
        if inference:
            return {"tokenized_signal": input_ids, "attn_mask": attn, "lengths": lengths}
        return {"tokenized_signal": input_ids, "attn_mask": attn,
                "quantized_signal_ids_input": labels, "position_ids": pos}

    def __call__(self, signal, percentiles, questions, answers):
        """signal `(B,12,L)` CUDA float64 -> training batch dict.  Only the first
        `pad_to_max` tokens of a record can ever be used (data_loader.py:106-107), and the
        encoder's output is prefix-stable, so the id buffer is capped there."""
        ids, counts = self.encode(signal, percentiles, max_tokens=self.pad_to_max)
        return self.assemble(ids, counts, questions, answers)


class ECGTokenDataset(torch.utils.data.Dataset):
    """Same constructor and item dict as the reference class (data_loader.py:34-132)."""

    def __init__(self, signal_path_list, text_path_list, vocab, merges, tokenizer=None, args=None):
        self.signal_path_list = np.array(signal_path_list)
        self.text_path_list = np.array(text_path_list)
        self.args = args
        self.vocab = vocab
        self.merges = merges
        self.tokenizer = tokenizer
        self.pad_id = tokenizer.convert_tokens_to_ids(tokenizer.pad_token)
        self.bos_id = tokenizer.convert_tokens_to_ids(tokenizer.bos_token)
        self.eos_id = tokenizer.convert_tokens_to_ids(tokenizer.eos_token)
        self.sig_start_id = tokenizer.convert_tokens_to_ids(["<sig_start>"])
        self.sig_end_id = tokenizer.convert_tokens_to_ids(["<sig_end>"])
        self.percentiles = np.load(args.percentiles, allow_pickle=True).item()
        keys = list(vocab.keys())
        lut = np.full(max(keys) + 1, self.pad_id, dtype=np.int32)
        lut[keys] = tokenizer.convert_tokens_to_ids([f"signal_{k}" for k in keys])   # data_loader.py:80
        self.assembler = BatchAssembler(merges, lut, self.pad_id, self.bos_id, self.eos_id,
                                        self.sig_start_id[0], self.sig_end_id[0], args.pad_to_max)

    def __len__(self):
        return len(self.signal_path_list)

    def _question_answer(self, text_label):
        ds = self.args.dataset
        if ds == "ptb_500":
            return "Could you please help me explain my ECG?", text_label
        if ds == "mimic_500":
            return text_label[0]["value"].replace("\n", "").replace("<ecg>", ""), text_label[1]["value"]
        if ds in ("ecg_qa_ptb_500", "ecg_qa_mimic_500", "ecg_qa_ptb_250", "ecg_qa_ptb_1250", "ecg_qa_ptb_2000"):
            answer = text_label[2]
            return text_label[1], (" ".join(answer) if isinstance(answer, list) else answer)
        raise KeyError(ds)

    def load_host(self, index):
        """The host half of __getitem__ (data_loader.py:55-80): read the .npy / .json pair, pick question and answer,
        tokenize them.  Returns (signal float64 (12, L), question ids, answer ids, question, answer) or None."""
        try:
            signal = np.load(self.signal_path_list[index])
            with open(self.text_path_list[index]) as f:
                text_label = json.load(f)
        except (FileNotFoundError, ValueError, OSError, KeyError) as e:
            print(f"Error loading files at index {index}: {e}")
            return None
        try:
            question, answer = self._question_answer(text_label)
            tq = self.tokenizer([question], return_tensors="np", add_special_tokens=False).input_ids[0].tolist()
            ta = self.tokenizer([answer], return_tensors="np", add_special_tokens=False).input_ids[0].tolist()
        except Exception as e:
            print(f"Error processing data at index {index}: {e}")
            return None
        return np.ascontiguousarray(signal, dtype=np.float64), tq, ta, question, answer

    def __getitem__(self, index):
        item = self.load_host(index)
        if item is None:
            return None
        signal, tq, ta, question, answer = item
        try:
            x = torch.from_numpy(signal[None]).cuda()
            inference = bool(self.args.inference)
            ids, counts = self.assembler.encode(x, self.percentiles,
                                                max_tokens=None if inference else self.args.pad_to_max)
        except Exception as e:
            print(f"Error processing data at index {index}: {e}")
            return None
        if inference:
            r = self.assembler.assemble(ids, counts, [tq], inference=True)
            n = int(r["lengths"][0].item())
            return {"answer": answer, "question": question,
                    "tokenized_signal": r["tokenized_signal"][0, :n].cpu(),
                    "attn_mask": r["attn_mask"][0, :n].cpu()}
        r = self.assembler.assemble(ids, counts, [tq], [ta])
        return {"tokenized_signal": r["tokenized_signal"][0].cpu(), "attn_mask": r["attn_mask"][0].cpu(),
                "quantized_signal_ids_input": r["quantized_signal_ids_input"][0].cpu(),
                "position_ids": r["position_ids"][0].cpu(), "signal": signal}


class DeviceBatchLoader:
    """Replaces `DataLoader(ECGTokenDataset, batch_size, shuffle, sampler, pin_memory=True)` (main.py:245-257, which
    runs with num_workers=0: every sample is read, quantised and encoded in the training process, one at a time).
    Here a thread pool reads and tokenizes the next batches while the current one trains; a batch's signals are
    stacked in pinned host memory, copied on a side stream, and quantise -> encode -> assemble run once per batch on
    the device.  Yields the reference's batch dict with device tensors (training: tokenized_signal, attn_mask,
    quantized_signal_ids_input, position_ids; inference, batch size 1: answer, question, tokenized_signal, attn_mask).
    Samples that fail to load are dropped from their batch; a batch with no valid sample is yielded as None, which
    the runners skip (train.py:17-19)."""

    def __init__(self, dataset: ECGTokenDataset, batch_size=1, shuffle=False, sampler=None, drop_last=False, seed=0,
                 workers=4, prefetch=2):
        self.dataset, self.batch_size, self.shuffle, self.sampler = dataset, int(batch_size), shuffle, sampler
        self.drop_last, self.seed, self.workers, self.prefetch = drop_last, seed, max(1, workers), max(1, prefetch)
        self.epoch = 0
        self.stall_s, self.batches_out = 0.0, 0       # host seconds the consumer spent waiting for a batch the readers had not finished; batches yielded
        if bool(dataset.args.inference) and self.batch_size != 1:
            raise ValueError("inference batches are single prompts (main.py:181-184 uses batch_size=1)")

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        if self.sampler is not None:
            return list(iter(self.sampler))
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n))
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        return torch.randperm(n, generator=g).tolist()

    def _host_batch(self, pool, idx):
        items = [it for it in pool.map(self.dataset.load_host, idx) if it is not None]
        if not items:
            return None
        sig = torch.from_numpy(np.stack([it[0] for it in items])).pin_memory()
        return sig, [it[1] for it in items], [it[2] for it in items], [it[3] for it in items], [it[4] for it in items]

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        order = self._order()
        self.epoch += 1
        batches = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        ds = self.dataset
        inference = bool(ds.args.inference)
        copy_stream = torch.cuda.Stream()
        with ThreadPoolExecutor(self.workers) as pool, ThreadPoolExecutor(1) as feeder:
            from collections import deque
            futures, staged = deque(), deque()
            nxt = 0

            def submit():
                nonlocal nxt
                if nxt < len(batches):
                    futures.append(feeder.submit(self._host_batch, pool, batches[nxt]))
                    nxt += 1

            def pump():                                       # one finished host batch -> pinned -> device (side stream)
                import time
                t0 = time.perf_counter()
                host = futures.popleft().result()
                self.stall_s += time.perf_counter() - t0
                submit()
                if host is None:
                    staged.append(None)
                    return
                sig, q, a, qs, ans = host
                with torch.cuda.stream(copy_stream):
                    dev = sig.cuda(non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                staged.append((dev, ev, q, a, qs, ans, sig))  # the pinned buffer stays alive until the batch is consumed

            for _ in range(self.prefetch):
                submit()
            while futures or staged:
                while futures and len(staged) < 2:            # the next batch's copy overlaps this batch's compute
                    pump()
                cur = staged.popleft()
                self.batches_out += 1
                if cur is None:
                    yield None
                    continue
                dev, ev, q, a, qs, ans, _pin = cur
                torch.cuda.current_stream().wait_event(ev)
                # `dev` was allocated on the copy stream but is read by kernels of the consumer's stream: tell the caching
                # allocator, or the block could be handed to the next batch's copy while the encoder still reads it
                dev.record_stream(torch.cuda.current_stream())
                if inference:
                    ids, counts = ds.assembler.encode(dev, ds.percentiles)
                    r = ds.assembler.assemble(ids, counts, q, inference=True)
                    n = int(r["lengths"][0].item())
                    yield {"answer": ans, "question": qs, "tokenized_signal": r["tokenized_signal"][:, :n],
                           "attn_mask": r["attn_mask"][:, :n]}
                else:
                    yield ds.assembler(dev, ds.percentiles, q, a)
