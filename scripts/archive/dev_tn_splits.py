"""Dev-only: the split weight-gradient shapes of the Llama-3.2-1B step (dW of qkv and o: fewer 256x256 tiles than CUs) against the number of
K-splits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
M = 32768
for N, K in [(3072, 2048), (2048, 2048), (2048, 8192), (16384, 2048)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    res = []
    for sp in (1, 2, 3, 4, 5, 6, 8):
        for _ in range(3): ops.gemm_tn(dy, x, splits=sp, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ops.gemm_tn(dy, x, splits=sp, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        res.append(f"s{sp}: {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.0f} TF")
    print(f"dW[{N},{K}]: " + "  ".join(res))
