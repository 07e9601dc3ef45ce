"""Dev-only: A/B of the 256x256 GEMM kernels (tile 256 = one-barrier-pair per K-tile, 258 = phased + staggered), interleaved."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd import decoder_ops as ops
shapes = [(32768, 3072, 2048), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 2048, 2048), (8192, 8192, 8192), (4096, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    outs = {t: torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for t in (256, 258)}
    best = {256: 1e9, 258: 1e9}
    for rep in range(6):
        for tile in (256, 258):
            ops.set_gemm_tile(tile)
            ops.gemm_nt(a, b, out=outs[tile])
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(4): ops.gemm_nt(a, b, out=outs[tile])
            torch.cuda.synchronize(); best[tile] = min(best[tile], (time.perf_counter() - t) / 4)
    ops.set_gemm_tile(0)
    same = torch.equal(outs[256], outs[258])
    ref = (a[:256].float() @ b[:512].float().T)
    err = (outs[258][:256, :512].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M{M} N{N} K{K}: tile256 {2*M*N*K/best[256]/1e12:.0f}  phased {2*M*N*K/best[258]/1e12:.0f} TFLOP/s   identical={same} relerr={err:.2e}")

for M, N, K in [(32768, 3072, 2048), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 2048, 2048), (4096, 132096, 2048)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    best = {256: 1e9, 258: 1e9}; outs = {}
    for rep in range(5):
        for tile in (258, 256):
            ops.set_gemm_tile(tile)
            outs[tile] = ops.gemm_tn(dy, x)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3): ops.gemm_tn(dy, x)
            torch.cuda.synchronize(); best[tile] = min(best[tile], (time.perf_counter() - t) / 3)
    ops.set_gemm_tile(0)
    d = (outs[256].float() - outs[258].float()).abs().max().item()
    print(f"TN dW[{N},{K}] over M={M}: unphased {2*M*N*K/best[258]/1e12:.0f}  phased {2*M*N*K/best[256]/1e12:.0f} TFLOP/s  maxdiff {d:.3g}")
