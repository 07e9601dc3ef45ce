"""Dev-only: fused quantise+encode time per record as the batch grows past one round of waves (4096 records = 16 per CU = one record per resident wave)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
for B in (2048, 4096, 6144, 8192, 12288, 16384):
    xd = torch.from_numpy(np.concatenate([base] * (B // 256))).cuda()
    for _ in range(3):
        tk.quantize_encode(xd, pc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        tk.quantize_encode(xd, pc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"B {B:6d}: {ms:.3f} ms  {ms / B * 4096:.3f} ms per 4096 records  {B * 60000 * 8 / ms / 1e6:.0f} GB/s of samples")
    del xd
