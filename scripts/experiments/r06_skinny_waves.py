import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ecg_byte_amd import decoder_ops as ops, _lib
L = _lib.lib()
L.ecgb_set_skinny_waves.argtypes = [ctypes.c_int]
H, I = 2048, 16384
g = torch.Generator(device="cuda").manual_seed(0)
Wgu = (torch.randn(2 * I, H, device="cuda", generator=g) * 0.02).bfloat16()
x = torch.randn(1, H, device="cuda", generator=g).bfloat16()
big = torch.zeros(1 << 28, dtype=torch.int32, device="cuda")
def timed(fn, reps=30):
    ts = []
    for _ in range(reps):
        big.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
ref = None
for wv in (4, 2, 1, 4, 1):
    L.ecgb_set_skinny_waves(wv)
    out = ops.gemm_nt_glu(x, Wgu, gelu_tanh=True, keep_gu=False)[1]
    if ref is None: ref = out.clone()
    print("waves per workgroup", wv, "glu gemv", round(timed(lambda: ops.gemm_nt_glu(x, Wgu, gelu_tanh=True, keep_gu=False)), 1), "us  same bits", torch.equal(out, ref))
