"""Dev-only: the four-wave kernel on the NN layout (dX = dY . W, W as nn.Linear stores it) against the eight-wave NN kernels: same bits, time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, N, K in [(512, 512, 128), (1024, 768, 320), (32768, 2048, 16384), (32768, 8192, 2048), (32768, 2048, 3072), (4096, 2048, 132096)]:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(K, N, device="cuda") * 0.05).to(torch.bfloat16)
    ref = ops.gemm_nn(a, b)
    got = ops.gemm_nn_w4(a, b)
    same = torch.equal(ref, got)
    if M >= 4096:
        res = {"w8": [], "w4": [], "blas": []}
        for rnd in range(3):
            res["w8"].append(timed(lambda: ops.gemm_nn(a, b)))
            res["w4"].append(timed(lambda: ops.gemm_nn_w4(a, b)))
            res["blas"].append(timed(lambda: torch.matmul(a, b)))
        fl = 2.0 * M * N * K
        print(f"NN M{M} N{N} K{K}: equal {same}  8-wave {min(res['w8']):.3f} ms {fl / min(res['w8']) / 1e9:.0f} TF/s   4-wave {min(res['w4']):.3f} ms {fl / min(res['w4']) / 1e9:.0f} TF/s   "
              f"hipBLASLt {min(res['blas']):.3f} ms {fl / min(res['blas']) / 1e9:.0f} TF/s", flush=True)
    else:
        print(f"NN M{M} N{N} K{K}: equal {same}  max diff {(ref.float() - got.float()).abs().max().item():.3g}", flush=True)

M, N, K = 32768, 2048, 16384
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = (torch.randn(K, N, device="cuda") * 0.05).to(torch.bfloat16)
for gm in (0, 2, 4, 8, 16):
    ops.set_gemm_w4_group_m(gm)
    print(f"NN M{M} N{N} K{K} tile order g{gm}: {min(timed(lambda: ops.gemm_nn_w4(a, b)) for _ in range(2)):.3f} ms")
ops.set_gemm_w4_group_m(8)

# TN: the weight gradients dW = dY^T . X
for Kc, M, N in [(128, 512, 512), (320, 768, 256), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 3072, 2048)]:
    torch.manual_seed(M + N + Kc)
    a = torch.randn(Kc, M, device="cuda").to(torch.bfloat16); b = (torch.randn(Kc, N, device="cuda") * 0.05).to(torch.bfloat16)
    ref = ops.gemm_tn(a, b, splits=1)
    got = ops.gemm_tn_w4(a, b)
    same = torch.equal(ref, got)
    if Kc >= 4096:
        res = {"w8": [], "w4": [], "blas": []}
        for rnd in range(3):
            res["w8"].append(timed(lambda: ops.gemm_tn(a, b, splits=1)))
            res["w4"].append(timed(lambda: ops.gemm_tn_w4(a, b)))
            res["blas"].append(timed(lambda: torch.matmul(a.T, b)))
        fl = 2.0 * M * N * Kc
        print(f"TN K{Kc} M{M} N{N}: equal {same}  8-wave {min(res['w8']):.3f} ms {fl / min(res['w8']) / 1e9:.0f} TF/s   4-wave {min(res['w4']):.3f} ms {fl / min(res['w4']) / 1e9:.0f} TF/s   "
              f"hipBLASLt {min(res['blas']):.3f} ms {fl / min(res['blas']) / 1e9:.0f} TF/s", flush=True)
    else:
        print(f"TN K{Kc} M{M} N{N}: equal {same}  max diff {(ref.float() - got.float()).abs().max().item():.3g}", flush=True)
