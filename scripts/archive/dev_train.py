"""Dev-only: HIP trainer at fixture scale (2000 records of 12xL, seed 1) vs the committed tokenizer."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
from helpers import load_tokenizer
import bench
from ecg_byte_amd.tokenizer import quantize
from ecg_byte_amd.trainer import bpe_train_device
from ecg_byte_amd import rust_bpe
tag = sys.argv[1] if len(sys.argv) > 1 else "c2"
L, nm = (5000, 4000) if tag == "c2" else (1000, 1000)
vocab, merges, pc = load_tokenizer(tag)
x = bench.make_signals(2000, L, seed=1, start=0, workers=int(_os.environ.get("ECGB_BENCH_WORKERS", "16")))   # 1 under rocprofv3: a fork pool under the profiler hangs
sym = quantize(torch.from_numpy(x).cuda(), pc).view(-1)
text = (sym + 97).contiguous()
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    ids, n_ids, pairs, n_done = bpe_train_device(text, nm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"train {text.numel()} symbols, {nm} merges: {dt:.3f} s, done {int(n_done)}, ids {int(n_ids)}")
v2, m2 = rust_bpe.vocab_merges_from_pairs(pairs[:int(n_done)].cpu().tolist())
print("merges equal committed fixture:", m2 == merges, " vocab equal:", v2 == vocab)
