"""Dev-only: time of one decode-attention call vs cache length (Gemma-2B head layout: 8 query heads, 1 kv head, head_dim 256)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecg_byte_amd import decoder_ops as ops
Hq, Hkv, D = 8, 1, 256
for B in (1, 8):
    cap = 4096
    cache = torch.randn(B, cap, 2 * Hkv * D, device="cuda").to(torch.bfloat16)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, cap, device="cuda")
    for L in (16, 64, 256, 728, 2048, 4096):
        for _ in range(10): ops.attn_decode(qkv, cache, mask, L, Hq, Hkv, D, 0.0625)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): ops.attn_decode(qkv, cache, mask, L, Hq, Hkv, D, 0.0625)
        e1.record(); torch.cuda.synchronize()
        one = e0.elapsed_time(e1) / 200 * 1e3
        ns = ops.decode_splits(L, B, Hq)
        sp = float("nan")
        if ns > 1:
            for _ in range(10): ops.attn_decode_split(qkv, cache, mask, L, Hq, Hkv, D, 0.0625, ns)
            torch.cuda.synchronize(); e0.record()
            for _ in range(200): ops.attn_decode_split(qkv, cache, mask, L, Hq, Hkv, D, 0.0625, ns)
            e1.record(); torch.cuda.synchronize()
            sp = e0.elapsed_time(e1) / 200 * 1e3
        print(f"B {B} len {L:5d}: one workgroup per head {one:7.1f} us   split over {ns:2d}: {sp:7.1f} us   (per call, incl. the q copy)")
