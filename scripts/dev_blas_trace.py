"""Dev-only: what hipBLASLt (through torch.matmul) runs for the C3 GEMM shapes -- kernel names (macro tile, MFMA shape, workgroup), registers, LDS --
next to this repo's kernels on the same tensors.  Run under `rocprofv3 --kernel-trace --stats`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops

REPS = int(os.environ.get("REPS", "30"))
shapes = [("NT", 32768, 16384, 2048), ("NT", 32768, 2048, 8192), ("NT", 32768, 3072, 2048), ("NT", 32768, 2048, 2048),
          ("NN", 32768, 2048, 16384), ("NN", 32768, 8192, 2048), ("TN", 16384, 2048, 32768), ("TN", 2048, 8192, 32768)]
for kind, M, N, K in shapes:
    if kind == "NT":      # C[M,N] = A[M,K] . B[N,K]^T
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        ours = lambda: ops.gemm_nt(a, b)
        blas = lambda: torch.matmul(a, b.T)
    elif kind == "NN":    # C[M,N] = A[M,K] . B[K,N]
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        ours = lambda: ops.gemm_nn(a, b)
        blas = lambda: torch.matmul(a, b)
    else:                 # C[M,N] = A[K,M]^T . B[K,N]
        a = torch.randn(K, M, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        ours = lambda: ops.gemm_tn(a, b)
        blas = lambda: torch.matmul(a.T, b)
    res = {}
    for name, fn in (("ours", ours), ("blas", blas), ("ours", ours), ("blas", blas)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS): fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / REPS)
    fl = 2.0 * M * N * K
    print(f"{kind} M{M} N{N} K{K}: ours {min(res['ours']):.3f} ms {fl / min(res['ours']) / 1e9:.0f} TF/s   hipBLASLt {min(res['blas']):.3f} ms {fl / min(res['blas']) / 1e9:.0f} TF/s", flush=True)
