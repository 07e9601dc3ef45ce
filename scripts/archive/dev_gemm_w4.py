"""Dev-only: the four-wave NT kernel (csrc/gemm_w4.hip) against the eight-wave kernels and hipBLASLt (torch.matmul) at the forward shapes of the C3 step, and its
tile order (blocks of g tile rows).  Same bits as the eight-wave kernels (same MFMA, same order over K).  The variants measured on the way (load order,
no barrier / no DMA wait / no stores, an LDS counter instead of s_barrier) are in scripts/experiments/r03_gemm_w4_variants.hip.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops

REPS = int(os.environ.get("REPS", "20"))
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS

for M, N, K in [(32768, 16384, 2048), (32768, 2048, 8192), (32768, 3072, 2048), (32768, 2048, 2048), (4096, 132096, 2048)]:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    ops.set_gemm_w4(False)
    ref = ops.gemm_nt(a, b)
    same = torch.equal(ref, ops.gemm_nt_w4(a, b))
    res = {"w8": [], "w4": [], "blas": []}
    for rnd in range(3):
        ops.set_gemm_w4(False)
        res["w8"].append(timed(lambda: ops.gemm_nt(a, b)))
        ops.set_gemm_w4(True)
        res["w4"].append(timed(lambda: ops.gemm_nt(a, b)))
        res["blas"].append(timed(lambda: torch.matmul(a, b.T)))
    fl = 2.0 * M * N * K
    grp = {}
    for gm in (0, 4, 8, 16, 32, 64):
        ops.set_gemm_w4_group_m(gm)
        assert torch.equal(ops.gemm_nt_w4(a, b), ref)
        grp[gm] = min(timed(lambda: ops.gemm_nt_w4(a, b)) for _ in range(2))
    ops.set_gemm_w4_group_m(8)
    print(f"M{M} N{N} K{K}: equal {same}  8-wave {min(res['w8']):.3f} ms {fl / min(res['w8']) / 1e9:.0f} TF/s   4-wave {min(res['w4']):.3f} ms {fl / min(res['w4']) / 1e9:.0f} TF/s   "
          f"hipBLASLt {min(res['blas']):.3f} ms {fl / min(res['blas']) / 1e9:.0f} TF/s     tile order (ms): " + "  ".join(f"g{k}: {t:.3f}" for k, t in grp.items()), flush=True)

# the gate|up projection with the GLU epilogue, and with a LoRA pair behind it
M, I, K = 32768, 8192, 2048
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); wgu = (torch.randn(2 * I, K, device="cuda") * 0.05).to(torch.bfloat16)
t = torch.randn(M, 64, device="cuda").to(torch.bfloat16); bl = (torch.randn(2 * I, 64, device="cuda") * 0.05).to(torch.bfloat16)
for name, kw in (("gate|up + GLU (SiLU)", {}), ("gate|up + GLU + LoRA pair", dict(a2=t, b2=bl)), ("gate|up + GLU (inference: h only)", dict(keep_gu=False))):
    res = {False: [], True: []}
    for rnd in range(3):
        for on in (False, True):
            ops.set_gemm_w4(on)
            res[on].append(timed(lambda: ops.gemm_nt_glu(x, wgu, **kw)))
    ops.set_gemm_w4(True)
    print(f"{name}: 8-wave {min(res[False]):.3f} ms   4-wave {min(res[True]):.3f} ms", flush=True)
