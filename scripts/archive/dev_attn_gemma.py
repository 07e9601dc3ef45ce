"""Dev-only: fused attention forward / backward at C5's shape (Gemma-2B: 8 query heads, 1 KV head of 256, S 2048, B 8) and C3's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecg_byte_amd import _lib
if os.environ.get("OLD"):
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_old.so")   # A/B against a second build
import torch
from ecg_byte_amd import decoder_ops as ops
for B, S, Hq, Hkv, D in [(8, 2048, 8, 1, 256), (32, 1024, 32, 8, 64), (8, 2048, 16, 4, 128)]:
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, S, device="cuda")
    sc = D ** -0.5
    o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, sc)
    do = torch.randn_like(o)
    for name, fn in (("fwd", lambda: ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, sc)), ("bwd", lambda: ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, sc))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        fl = 4 * S * S * D / 2 * B * Hq * (1 if name == "fwd" else 2.5)
        print(f"B{B} S{S} {Hq}/{Hkv}x{D} {name}: {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TFLOP/s (causal-counted)")
