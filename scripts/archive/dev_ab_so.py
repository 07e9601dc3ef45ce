import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_byte_amd import _lib
if os.environ.get("OLD"):
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_old.so")
import torch
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
scale = 0.125
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16); do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
r = [timed(lambda: ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale)) for _ in range(4)]
print("OLD" if os.environ.get("OLD") else "NEW", "bwd %.3f ms (min) %.3f (median)" % (min(r), sorted(r)[2]))
