"""Dev-only: time the fused attention kernels at the Llama-3.2-1B shape (B=32, S=1024, 32/8 heads, D=64)."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time, math
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
QKV = Hq * D + 2 * Hkv * D
qkv = torch.randn(B * S, QKV, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda"); mask[:, :100] = 0
do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
scale = 1 / math.sqrt(D)
for _ in range(2):
    o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale); d = ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(5): d = ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale)
torch.cuda.synchronize(); t2 = time.perf_counter()
fl = 2 * 2 * B * Hq * S * S * D / 2   # causal half of QK^T + PV
print(f"fwd {(t1-t)/5*1e3:.3f} ms ({fl/((t1-t)/5)/1e12:.0f} TF/s causal-algorithmic)  bwd {(t2-t1)/5*1e3:.3f} ms ({2.5*fl/((t2-t1)/5)/1e12:.0f} TF/s)")
