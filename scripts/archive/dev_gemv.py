"""Dev-only: the few-row GEMM (decode step) at the Gemma-2B weight shapes: time and bytes/s of reading the weight once."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecg_byte_amd import decoder_ops as ops
shapes = {"qkv": (2560, 2048), "o": (2048, 2048), "gate_up": (32768, 2048), "down": (2048, 16384), "lm_head": (259762, 2048)}
for M in (1, 8):
    for name, (N, K) in shapes.items():
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(5): ops.gemm_nt(x, w, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): ops.gemm_nt(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"M {M} {name:8s} N {N:6d} K {K:5d}: {us:7.1f} us  {N*K*2/us/1e6:6.2f} TB/s")
