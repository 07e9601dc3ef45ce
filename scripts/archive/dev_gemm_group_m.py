"""Dev-only A/B: the GEMM kernels of the C3 train step with the tile order row by row (group_m 0, round 2) against blocks of 4 / 8 / 16 tile rows
(ecgb_set_gemm_group_m), same tensors, one process, alternating rounds; HIP events.  Same bits whatever the order (checked)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops

T, H, I, QKV = 32768, 2048, 8192, 3072
bf = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)


def timed(fn, reps=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x, wgu, wd, wqkv, wo = bf(T, H), bf(2 * I, H), bf(H, I), bf(QKV, H), bf(H, H)
hm, dy, dgu, dqkv = bf(T, I), bf(T, H), bf(T, 2 * I), bf(T, QKV)
gu = bf(T, 2 * I)
cases = [
    ("NT qkv        [T,2048]x[3072,2048]^T", lambda: ops.gemm_nt(x, wqkv), 2 * T * QKV * H),
    ("NT o          [T,2048]x[2048,2048]^T", lambda: ops.gemm_nt(x, wo), 2 * T * H * H),
    ("NT gate|up+GLU [T,2048]x[16384,2048]^T", lambda: ops.gemm_nt_glu(x, wgu), 2 * T * 2 * I * H),
    ("NT down       [T,8192]x[2048,8192]^T", lambda: ops.gemm_nt(hm, wd), 2 * T * H * I),
    ("NN dX gate|up [T,16384]x[16384,2048]", lambda: ops.gemm_nn(dgu, wgu), 2 * T * 2 * I * H),
    ("NN dX down+GLUbwd [T,2048]x[2048,8192]", lambda: ops.gemm_nn_glu_bwd(dy, wd, gu), 2 * T * H * I),
    ("NN dX qkv     [T,3072]x[3072,2048]", lambda: ops.gemm_nn(dqkv, wqkv), 2 * T * QKV * H),
    ("NN dX o       [T,2048]x[2048,2048]", lambda: ops.gemm_nn(dy, wo), 2 * T * H * H),
    ("TN dW gate|up [T,16384]^T x [T,2048]", lambda: ops.gemm_tn(dgu, x), 2 * T * 2 * I * H),
    ("TN dW down    [T,2048]^T x [T,8192]", lambda: ops.gemm_tn(dy, hm), 2 * T * H * I),
    ("TN dW qkv     [T,3072]^T x [T,2048]", lambda: ops.gemm_tn(dqkv, x), 2 * T * QKV * H),
    ("TN dW o       [T,2048]^T x [T,2048]", lambda: ops.gemm_tn(dy, x), 2 * T * H * H),
]
groups = (0, 4, 8, 16)
tot = {g: 0.0 for g in groups}
for name, fn, flops in cases:
    res = {g: [] for g in groups}
    ref = None
    for rnd in range(3):
        for g in groups:
            ops.set_gemm_group_m(g)
            out = fn()
            out = out[-1] if isinstance(out, tuple) else out
            if ref is None:
                ref = out.clone()
            elif rnd == 0:
                assert torch.equal(out, ref), (name, g)
            fn(); torch.cuda.synchronize()
            res[g].append(timed(fn))
    per_layer = {g: min(v) for g, v in res.items()}
    for g in groups:
        tot[g] += per_layer[g]
    print(f"{name:42s} " + "  ".join(f"g{g}: {per_layer[g]:.3f} ms {flops / per_layer[g] / 1e9:5.0f} TF/s" for g in groups))
print("sum per layer: " + "  ".join(f"g{g}: {tot[g]:.3f} ms" for g in groups))
ops.set_gemm_group_m(0)
