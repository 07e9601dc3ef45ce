"""Dev-only: a few launches of the fused quantise+encode on the bench workload (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
xd = torch.from_numpy(np.concatenate([base] * (B // 256))).cuda()
for _ in range(4):
    tk.quantize_encode(xd, pc)
torch.cuda.synchronize()
