"""Dev-only: the dispatch threshold of the four-wave kernel (K-tiles per workgroup) on the shapes between 128 and 256: qkv / o forward (NT) and their input gradients (NN)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
REPS = 30
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
cases = []
for M, N, K in [(32768, 3072, 2048), (32768, 2048, 2048)]:
    a, b = bf(M, K), bf(N, K, sc=0.05)
    cases.append((f"NT [{M}, {K}] -> {N}", (lambda a=a, b=b: ops.gemm_nt(a, b)), 2.0 * M * N * K))
    t, bl = bf(M, 64), bf(N, 64, sc=0.05)
    cases.append((f"NT [{M}, {K}] -> {N} + LoRA pair", (lambda a=a, b=b, t=t, bl=bl: ops.gemm_nt(a, b, a2=t, b2=bl)), 2.0 * M * N * (K + 64)))
for M, N, K in [(32768, 2048, 3072), (32768, 2048, 2048)]:
    a, b = bf(M, K), bf(K, N, sc=0.05)
    cases.append((f"NN [{M}, {K}] . [{K}, {N}]", (lambda a=a, b=b: ops.gemm_nn(a, b)), 2.0 * M * N * K))
for name, fn, fl in cases:
    res = {256: [], 128: []}
    outs = {}
    for rnd in range(3):
        for th in (256, 128):
            ops.set_gemm_w4_min_ktiles(th)
            outs[th] = fn().clone()
            res[th].append(timed(fn))
    ops.set_gemm_w4_min_ktiles(128)
    print(f"{name}: same bits {torch.equal(outs[256], outs[128])}   eight-wave {min(res[256]):.3f} ms {fl / min(res[256]) / 1e9:.0f} TF/s   four-wave {min(res[128]):.3f} ms {fl / min(res[128]) / 1e9:.0f} TF/s ({(min(res[128]) / min(res[256]) - 1) * 100:+.1f} %)", flush=True)
