"""Dev-only: input-gradient products of the Llama-3.2-1B step, NT on a transposed shadow against the NN kernel on the weight as stored."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
M = 32768
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for name, out_f, in_f in (("qkv", 3072, 2048), ("o", 2048, 2048), ("gu", 16384, 2048), ("down", 2048, 8192)):
    dy = torch.randn(M, out_f, device="cuda").to(torch.bfloat16); w = (torch.randn(out_f, in_f, device="cuda") * 0.02).to(torch.bfloat16)
    wt = ops.transpose(w)
    t0 = timed(lambda: ops.gemm_nt(dy, wt)); t1 = timed(lambda: ops.gemm_nn(dy, w))
    fl = 2 * M * out_f * in_f
    print(f"dX {name}: NT on shadow {t0:.3f} ms ({fl / t0 / 1e9:.0f} TF)   NN {t1:.3f} ms ({fl / t1 / 1e9:.0f} TF)")
