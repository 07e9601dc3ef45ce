"""Dev-only: TWO builds of the library in one process (the in-tree one and ECGB_SO_B, default libecgbyte_hip_old.so) on the C3 step's four-wave GEMM shapes, interleaved:
the only way to see a 1-3 % change of an MFMA kernel on this pool.  Same bits asserted."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
A = _lib.lib()
B = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), os.environ.get("ECGB_SO_B", "libecgbyte_hip_old.so")))
vp, ll, ci, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_float
for L in (A, B):
    for fn in ("ecgb_gemm_nt_w4_bf16", "ecgb_gemm_nn_w4_bf16", "ecgb_gemm_tn_w4_bf16"):
        getattr(L, fn).argtypes = [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, vp]; getattr(L, fn).restype = ci
REPS = int(os.environ.get("REPS", "20"))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS
def ab(name, call, out, flops):
    res = {0: [], 1: []}
    outs = []
    for i, L in enumerate((A, B)):
        out.zero_(); assert call(L) == 0; outs.append(out.clone())
    for rnd in range(4):
        for i, L in enumerate((A, B)): res[i].append(timed(lambda: call(L)))
    ta, tb = min(res[0]), min(res[1])
    print(f"{name}: same bits {torch.equal(outs[0], outs[1])}   new {ta:.4f} ms {flops / ta / 1e9:.0f} TF/s   old {tb:.4f} ms ({(ta / tb - 1) * 100:+.2f} %)", flush=True)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
torch.manual_seed(0)
for M, N, K in [(32768, 16384, 2048), (32768, 2048, 8192), (32768, 3072, 2048), (32768, 2048, 2048)]:
    a, b, c = bf(M, K), bf(N, K), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ab(f"NT [{M}, {K}] -> {N}", lambda L: L.ecgb_gemm_nt_w4_bf16(p(a), K, p(b), K, p(c), N, M, N, K, 1.0, st()), c, 2.0 * M * N * K)
for M, N, K in [(32768, 2048, 16384), (32768, 8192, 2048)]:
    a, b, c = bf(M, K), bf(K, N, sc=0.1), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ab(f"NN [{M}, {K}] . [{K}, {N}]", lambda L: L.ecgb_gemm_nn_w4_bf16(p(a), K, p(b), N, p(c), N, M, N, K, 1.0, st()), c, 2.0 * M * N * K)
for Kc, M, N in [(32768, 16384, 2048), (32768, 2048, 8192)]:
    a, b, c = bf(Kc, M), bf(Kc, N, sc=0.1), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ab(f"TN [{Kc}, {M}]^T . [{Kc}, {N}]", lambda L: L.ecgb_gemm_tn_w4_bf16(p(a), M, p(b), N, p(c), N, M, N, Kc, 1.0, st()), c, 2.0 * M * N * Kc)
