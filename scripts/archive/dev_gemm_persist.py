"""Dev-only: the persistent tile loop of the 256x256 NT kernel (gemm_nt_kernel_m16pp) against the one-tile-per-workgroup kernel (set_gemm_tile(259)): same bits?  time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops


def timed(f, n=10):
    best = 1e9
    for _ in range(3):
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best * 1e3


for M, N, K in [(32768, 3072, 2048), (32768, 2048, 2048), (32768, 2048, 8192), (32768, 16384, 2048), (4096, 132608, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    a2 = torch.randn(M, 64, device="cuda").to(torch.bfloat16); b2 = (torch.randn(N, 64, device="cuda") * 0.1).to(torch.bfloat16)
    for cat in (False, True):
        kw = dict(a2=a2, b2=b2) if cat else {}
        ops.set_gemm_tile(259); want = ops.gemm_nt(a, b, **kw); t0 = timed(lambda: ops.gemm_nt(a, b, **kw))
        ops.set_gemm_tile(0); got = ops.gemm_nt(a, b, **kw); t1 = timed(lambda: ops.gemm_nt(a, b, **kw))
        same = all(torch.equal(ops.gemm_nt(a, b, **kw), want) for _ in range(3)) and torch.equal(got, want)
        print(f"NT M{M} N{N} K{K} cat={cat}: one tile {t0:.3f} ms ({2*M*N*K/t0/1e9:.0f} TFLOP/s)  persistent {t1:.3f} ms ({2*M*N*K/t1/1e9:.0f})  same bits {same}")
for name, M, I, K, gelu in (("llama", 32768, 8192, 2048, False), ("gemma", 16384, 16384, 2048, True)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(2 * I, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    a2 = torch.randn(M, 64, device="cuda").to(torch.bfloat16); b2 = (torch.randn(2 * I, 64, device="cuda") * 0.1).to(torch.bfloat16)
    for cat in (False, True):
        kw = dict(a2=a2, b2=b2) if cat else {}
        ops.set_gemm_tile(259); w_gu, w_h = ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw); t0 = timed(lambda: ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw))
        ops.set_gemm_tile(0); g_gu, g_h = ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw); t1 = timed(lambda: ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw))
        same = torch.equal(g_gu, w_gu) and torch.equal(g_h, w_h)
        for _ in range(3):
            r_gu, r_h = ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw); same = same and torch.equal(r_gu, w_gu) and torch.equal(r_h, w_h)
        print(f"GLU {name} cat={cat}: one tile {t0:.3f} ms  persistent {t1:.3f} ms  same bits {same}")
for M, N, K in [(32768, 2048, 3072), (32768, 2048, 2048), (32768, 2048, 16384), (32768, 8192, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(K, N, device="cuda") * K ** -0.5).to(torch.bfloat16)
    ops.set_gemm_tile(259); want = ops.gemm_nn(a, b); t0 = timed(lambda: ops.gemm_nn(a, b))
    ops.set_gemm_tile(0); got = ops.gemm_nn(a, b); t1 = timed(lambda: ops.gemm_nn(a, b))
    same = all(torch.equal(ops.gemm_nn(a, b), want) for _ in range(3)) and torch.equal(got, want)
    print(f"NN M{M} N{N} K{K}: one tile {t0:.3f} ms ({2*M*N*K/t0/1e9:.0f} TFLOP/s)  persistent {t1:.3f} ms ({2*M*N*K/t1/1e9:.0f})  same bits {same}")
M, I, K = 32768, 8192, 2048
dy = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(K, I, device="cuda") * K ** -0.5).to(torch.bfloat16); gu = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16)
ops.set_gemm_tile(259); want = ops.gemm_nn_glu_bwd(dy, w, gu); t0 = timed(lambda: ops.gemm_nn_glu_bwd(dy, w, gu))
ops.set_gemm_tile(0); got = ops.gemm_nn_glu_bwd(dy, w, gu); t1 = timed(lambda: ops.gemm_nn_glu_bwd(dy, w, gu))
print(f"NN + GLU backward: one tile {t0:.3f} ms  persistent {t1:.3f} ms  same bits {torch.equal(got, want) and all(torch.equal(ops.gemm_nn_glu_bwd(dy, w, gu), want) for _ in range(3))}")
