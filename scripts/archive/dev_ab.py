"""Dev-only: time the encode kernels under each plan mode at bench size."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
x = np.concatenate([base] * (B // 256))
xd = torch.from_numpy(x).cuda()
ids = torch.empty((B, 60000), dtype=torch.int32, device="cuda"); counts = torch.empty((B,), dtype=torch.int32, device="cuda")
ref = None
for mode in [int(a) for a in sys.argv[2:]] or [1, 2, 3]:
    set_encode_plan(mode)
    for _ in range(2): tk.quantize_encode(xd, pc, out=(ids, counts))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): tk.quantize_encode(xd, pc, out=(ids, counts))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    c = counts.clone()
    if ref is None: ref = (ids[:, :6000].clone(), c)
    same = bool(torch.equal(c, ref[1]) and torch.equal(ids[:, :6000] * (torch.arange(6000, device="cuda")[None] < c[:, None]), ref[0] * (torch.arange(6000, device="cuda")[None] < c[:, None])))
    print(f"mode {mode}: {dt*1e3:.3f} ms  {B*480000/dt/1e9:.0f} GB/s  same={same}")
