"""Dev-only A/B: the NT / TN GEMM of the current library against the round-1 build of gemm.hip (scripts/experiments/libgemm_r01.so, built
by hand from git history; not shipped) on the same tensors in one process, alternating."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd import _lib
new = _lib.lib()
old = C.CDLL(os.path.join(ROOT, "scripts", "experiments", "libgemm_r01.so"))
vp, ll, ci, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_float
old.ecgb_gemm_nt_bf16.argtypes = [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, ci, ll, ll, ll, vp]
old.ecgb_gemm_tn_bf16.argtypes = [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, vp]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for M, N, K in [(32768, 3072, 2048), (32768, 16384, 2048), (32768, 2048, 8192), (32768, 2048, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for name, L in (("r01", old), ("now", new)):
            for _ in range(3): L.ecgb_gemm_nt_bf16(p(a), K, p(b), K, p(out), N, M, N, K, 1.0, 0, 1, 0, 0, 0, st())
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): L.ecgb_gemm_nt_bf16(p(a), K, p(b), K, p(out), N, M, N, K, 1.0, 0, 1, 0, 0, 0, st())
            torch.cuda.synchronize(); res.setdefault(name, []).append((time.perf_counter() - t) / 10)
    print(f"NT M{M} N{N} K{K}: " + "  ".join(f"{n} {2*M*N*K/min(v)/1e12:.0f} TFLOP/s" for n, v in res.items()))
