"""Dev-only: encode time per launch over batch sizes (B in {1, 64, 1024, 4096}) -- which kernel the plan picks and what it costs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer
for tag, L in (("c1", 1000), ("c2", 5000)):
    _, merges, pc = load_tokenizer(tag)
    tk = HipTokenizer(merges)
    base = synth.synth_ecg(256, L, seed=0)
    for B in (1, 4, 64, 256, 512, 1024, 2048, 4096, 8192):
        x = np.concatenate([base] * max(1, B // 256))[:B]
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        for _ in range(3): ids, counts = tk.quantize_encode(xd, pc)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for _ in range(n): ids, counts = tk.quantize_encode(xd, pc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        toks = int(counts.sum().item())
        print(f"{tag} L {L} B {B:5d}: {dt*1e6:9.1f} us/launch  {B/dt/1e6:7.3f} M records/s  {toks/dt/1e9:7.2f} G tokens/s  {B*12*L*8/dt/1e9:8.1f} GB/s in")
