"""Dev-only: the LoRA passes at the Llama-3.2-1B step shapes (T = 32 768): time and effective HBM rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
T = 32768


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


for name, K, n_sub, n_fields in (("o", 2048, 1, 1), ("down", 8192, 1, 1), ("qkv", 2048, 3, 3), ("gu", 2048, 2, 2)):
    x = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    A = (torch.randn(64, K, device="cuda") * 0.02).to(torch.bfloat16)
    At = A.T.contiguous()
    dt = torch.randn(T, 64, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.05):
        t0 = timed(lambda: ops.lora_down(x, A, n_sub, n_fields, 2.0, p, 1234))
        by = T * K * 2
        dx = torch.zeros(T, K, device="cuda", dtype=torch.bfloat16)
        t1 = timed(lambda: ops.lora_dx_(dx, dt, At, n_sub, n_fields, 2.0, p, 1234))
        da = torch.zeros(16 * n_sub, K, device="cuda", dtype=torch.bfloat16)
        t2 = timed(lambda: ops.lora_da(x, dt, da, n_sub, n_fields, 2.0, p, 1234))
        w = 16 * n_sub // n_fields
        t3 = timed(lambda: [ops.gemm_tn(dt[:, w * f: w * f + w], x, alpha=2.0, out=da[w * f: w * f + w]) for f in range(n_fields)])   # the TN GEMMs on masked copies it replaces (x stands in for xd)
        print(f"{name:5s} p={p}: lora_down {t0:.3f} ms ({by / t0 / 1e9:.2f} TB/s)   lora_dx {t1:.3f} ms ({T * K * 4 / t1 / 1e9:.2f} TB/s)"
              f"   lora_da {t2:.3f} ms ({by / t2 / 1e9:.2f} TB/s; per-module TN GEMMs {t3:.3f} ms)")
