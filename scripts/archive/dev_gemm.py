"""Dev-only: TFLOP/s of ecgb_gemm_nt_bf16 at the Llama-3.2-1B projection shapes (random data)."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd import decoder_ops as ops
shapes = [(32768, 3072, 2048), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 2048, 2048), (4096, 132096, 2048), (2048, 8192, 32768)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for tile in (128, 256, 257):
        ops.set_gemm_tile(tile)
        for _ in range(2): ops.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ops.gemm_nt(a, b, out=out)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 5)
    ops.set_gemm_tile(0)
    dt = min(res)
    t0 = time.perf_counter()
    for _ in range(5): torch.matmul(a, b.T, out=out)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 5
    print(f"M{M} N{N} K{K}: tile128 {2*M*N*K/res[0]/1e12:.0f}  tile256 {2*M*N*K/res[1]/1e12:.0f}  tile256/m16 {2*M*N*K/res[2]/1e12:.0f} TFLOP/s   (torch/hipBLASLt {2*M*N*K/dt2/1e12:.0f})")

for M, N, K in [(32768, 3072, 2048), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 2048, 2048)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    for _ in range(2): ops.gemm_tn(dy, x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.gemm_tn(dy, x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"TN dW[{N},{K}] over M={M}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.0f} TFLOP/s")
