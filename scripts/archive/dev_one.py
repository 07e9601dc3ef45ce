"""Dev-only: a few launches of the fused kernel at bench size (for rocprofv3 counter passes)."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
x = np.concatenate([base] * (B // 256))
xd = torch.from_numpy(x).cuda()
for _ in range(3):
    tk.quantize_encode(xd, pc)
torch.cuda.synchronize()
