"""Dev-only: where a 256x256 tile of gemm_nt_kernel_m16p spends its time (prologue until the first K-tile has landed, K loop, epilogue), from the
-DECGB_PROFILE build (make -C ecg_byte_amd/csrc prof): shader cycles of wave 0, averaged over the tiles of a launch."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
from ecg_byte_amd import decoder_ops as ops
L = _lib.lib()
L.ecgb_debug_gemm_profile.argtypes = [C.c_void_p, C.c_int]
for M, N, K in [(32768, 16384, 2048), (32768, 3072, 2048), (32768, 2048, 8192), (32768, 2048, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    L.ecgb_debug_gemm_profile(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(a, b); e1.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 12)()
    L.ecgb_debug_gemm_profile(out, 0)
    n = out[3]
    print(f"NT M{M} N{N} K{K}: {e0.elapsed_time(e1):.3f} ms, {n} tiles, cycles per tile: prologue {out[0] / n:.0f}  K loop {out[1] / n:.0f} ({out[1] / n / (K // 64):.0f} per K-tile)  "
          f"epilogue {out[2] / n:.0f}")
for M, N, K in [(32768, 8192, 2048), (32768, 2048, 3072), (32768, 2048, 16384), (32768, 2048, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    for _ in range(3): ops.gemm_nn(a, b)
    torch.cuda.synchronize()
    L.ecgb_debug_gemm_profile(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nn(a, b); e1.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 12)()
    L.ecgb_debug_gemm_profile(out, 0)
    n = out[7]
    print(f"NN M{M} N{N} K{K}: {e0.elapsed_time(e1):.3f} ms ({2 * M * N * K / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s with the timers in), {n} tiles, cycles per tile: prologue {out[4] / n:.0f}  "
          f"K loop {out[5] / n:.0f} ({out[5] / n / (K // 64):.0f} per K-tile)  epilogue {out[6] / n:.0f}")
for M, N, K in [(32768, 16384, 2048), (32768, 2048, 8192), (32768, 3072, 2048), (32768, 2048, 2048)]:       # dW [N, K] = dY[M, N]^T . X[M, K]
    a = torch.randn(M, N, device="cuda").to(torch.bfloat16); b = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    for _ in range(3): ops.gemm_tn(a, b)
    torch.cuda.synchronize()
    L.ecgb_debug_gemm_profile(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_tn(a, b); e1.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 12)()
    L.ecgb_debug_gemm_profile(out, 0)
    wgs = ((N + 255) // 256) * ((K + 255) // 256) * ops.tn_splits(N, K, M)
    print(f"TN dW[{N},{K}] over M{M}: {e0.elapsed_time(e1):.3f} ms incl. the slab sum ({2 * M * N * K / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s with the timers in), {wgs} workgroups, "
          f"cycles per workgroup: prologue {out[8] / wgs:.0f}  K loop {out[9] / wgs:.0f} ({out[9] / max(out[11], 1):.0f} per K-tile)  epilogue {out[10] / wgs:.0f}")
