"""Dev-only: encode time per launch under the launch plans (0 auto = 16 waves/CU, 3 = 8 waves/CU with long segments)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
for B in (2048, 4096, 8192):
    xd = torch.from_numpy(np.concatenate([base] * (B // 256))).cuda()
    ref = None
    for plan in (0, 3, 0, 3):
        set_encode_plan(plan)
        ids, counts = tk.quantize_encode(xd, pc)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ids, counts = tk.quantize_encode(xd, pc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        if ref is None: ref = (ids.clone(), counts.clone())
        same = torch.equal(counts, ref[1]) and bool(((ids == ref[0]) | (torch.arange(ids.shape[1], device="cuda")[None] >= counts[:, None])).all())
        print(f"B {B} plan {plan}: {dt*1e3:.3f} ms  ({B/dt/1e6:.2f} M records/s)  same={same}")
set_encode_plan(0)
