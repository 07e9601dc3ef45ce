"""Dev-only: the down projection's input gradient with the GLU backward in the epilogue, eight-wave kernel against the four-wave kernel (round 4), C3's shape."""
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
M, I, K = 32768, 8192, 2048
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
dy, w, gu = bf(M, K), bf(K, I, sc=K ** -0.5), bf(M, 2 * I)
res = {0: [], 1: []}; outs = {}
for rnd in range(3):
    for on in (0, 1):
        ops.set_gemm_w4(bool(on))
        outs[on] = ops.gemm_nn_glu_bwd(dy, w, gu).clone()
        res[on].append(timed(lambda: ops.gemm_nn_glu_bwd(dy, w, gu)))
ops.set_gemm_w4(True)
t_nn = min(timed(lambda: ops.gemm_nn(dy, w)) for _ in range(3))
print(f"dX down + GLU backward [{M}, {K}] . [{K}, {I}]: same bits {torch.equal(outs[0], outs[1])}   eight-wave {min(res[0]):.3f} ms   four-wave {min(res[1]):.3f} ms   (plain NN product on four waves {t_nn:.3f} ms)")
