"""Dev-only: gate|up projection + GLU as two launches against the GLU-epilogue GEMM, Llama-3.2-1B and Gemma-2B step shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


for name, M, I, K, gelu in (("llama", 32768, 8192, 2048, False), ("gemma", 16384, 16384, 2048, True)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(2 * I, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    a2 = torch.randn(M, 64, device="cuda").to(torch.bfloat16); b2 = (torch.randn(2 * I, 64, device="cuda") * 0.1).to(torch.bfloat16)
    for cat in (False, True):
        kw = dict(a2=a2, b2=b2) if cat else {}
        t0 = timed(lambda: ops.glu_fwd(ops.gemm_nt(a, b, **kw), gelu_tanh=gelu))
        t1 = timed(lambda: ops.gemm_nt_glu(a, b, gelu_tanh=gelu, **kw))
        t2 = timed(lambda: ops.gemm_nt_glu(a, b, gelu_tanh=gelu, keep_gu=False, **kw))
        tg = timed(lambda: ops.gemm_nt(a, b, **kw))
        print(f"{name} cat={cat}: gemm {tg:.3f}  gemm+glu {t0:.3f}  fused {t1:.3f}  fused, no gate|up {t2:.3f} ms")
