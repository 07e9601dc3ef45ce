"""Dev: the fused quantise + encode launch on the bench batch under each plan (0 auto = lane per chunk where it applies, 5 = segment kernels), HIP events.
    python scripts/dev_encode_plans.py [records] [plans, e.g. 5,4]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plans = [int(p) for p in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5, 4]
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
x = synth.synth_ecg(min(B, 4096), 5000, seed=0)
xd = torch.from_numpy(x).cuda()
if B > 4096:
    xd = xd.repeat((B + 4095) // 4096, 1, 1)[:B].contiguous()
n = 60000
ids = torch.empty((B, n), dtype=torch.int32, device="cuda"); counts = torch.empty((B,), dtype=torch.int32, device="cuda")
res = {}
for p in plans:
    set_encode_plan(p)
    for _ in range(3):
        tk.quantize_encode(xd, pc, ids_stride=n, out=(ids, counts))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        tk.quantize_encode(xd, pc, ids_stride=n, out=(ids, counts))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    toks = int(counts.sum().item())
    alg = B * 8 * n + 4 * toks
    res[p] = (ids.clone(), counts.clone())
    print(f"plan {p}: {ms:.4f} ms per launch of {B} records, {toks / ms / 1e6:.2f} G tokens/s, {alg / ms / 1e6 / 8000:.3f} of the HBM roofline", flush=True)
set_encode_plan(0)
if len(res) > 1:
    (i0, c0), (i1, c1) = list(res.values())[:2]
    valid = torch.arange(n, device="cuda")[None, :] < c0[:, None]
    print("same counts:", bool(torch.equal(c0, c1)), " same ids:", bool(((i0 == i1) | ~valid).all()))
