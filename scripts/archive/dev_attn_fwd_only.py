"""Dev-only: time the head_dim-64 forward attention kernel alone at the C3 shape; ECGB_SO picks a (diagnostic) build of the library."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
if os.environ.get("ECGB_SO"): _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
f = lambda: ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
best = 1e9
for rnd in range(4):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"forward {best * 1e3:.1f} us")
