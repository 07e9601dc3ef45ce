"""Dev-only: the two K-tile schedules of the four-wave GEMM kernel (csrc/gemm_w4.hip, `ecgb_set_gemm_w4_sched`) interleaved in one process on the C3 step's
shapes -- 0: one rendezvous per K-tile (round 3), 1: four barriers behind counted waits (round 4) -- with hipBLASLt (torch.matmul) beside them as the yardstick.
Same bits under both schedules (asserted)."""
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


def ab(name, fn, flops, blas=None):
    outs, res = {}, {0: [], 1: [], "blas": []}
    for s in (0, 1):
        ops.set_gemm_w4_sched(s)
        o = fn()
        outs[s] = [t.clone() for t in (o if isinstance(o, (tuple, list)) else (o,)) if t is not None]
    same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
    for rnd in range(3):
        for s in (0, 1):
            ops.set_gemm_w4_sched(s)
            res[s].append(timed(fn))
        if blas is not None: res["blas"].append(timed(blas))
    ops.set_gemm_w4_sched(1)
    t0, t1 = min(res[0]), min(res[1])
    line = f"{name}: same bits {same}   sched 0 {t0:.3f} ms {flops / t0 / 1e9:.0f} TF/s   sched 1 {t1:.3f} ms {flops / t1 / 1e9:.0f} TF/s ({(t1 / t0 - 1) * 100:+.1f} %)"
    if blas is not None: line += f"   hipBLASLt {min(res['blas']):.3f} ms {flops / min(res['blas']) / 1e9:.0f} TF/s"
    print(line, flush=True)
    assert same, name


torch.manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
ops.set_gemm_w4(True)
# NT: gate|up, down, the loss head, qkv and o by name (the dispatch keeps those two on eight waves)
for M, N, K in [(32768, 16384, 2048), (32768, 2048, 8192), (4096, 132096, 2048), (32768, 3072, 2048), (32768, 2048, 2048)]:
    a, b = bf(M, K), bf(N, K)
    ab(f"NT [{M}, {K}] -> {N}", lambda: ops.gemm_nt_w4(a, b), 2.0 * M * N * K, lambda: torch.matmul(a, b.T))
    del a, b
# NN: dX of gate|up (K 16384), dX of down (K 2048, by name)
for M, N, K in [(32768, 2048, 16384), (32768, 8192, 2048)]:
    a, b = bf(M, K), bf(K, N, sc=0.1)
    ab(f"NN [{M}, {K}] . [{K}, {N}]", lambda: ops.gemm_nn_w4(a, b), 2.0 * M * N * K, lambda: torch.matmul(a, b))
    del a, b
# TN: dW of gate|up and of down
for Kc, M, N in [(32768, 16384, 2048), (32768, 2048, 8192)]:
    a, b = bf(Kc, M), bf(Kc, N, sc=0.1)
    ab(f"TN [{Kc}, {M}]^T . [{Kc}, {N}]", lambda: ops.gemm_tn_w4(a, b), 2.0 * M * N * Kc, lambda: torch.matmul(a.T, b))
    del a, b
# the gate|up projection with the GLU epilogue, plain and with a LoRA pair (the pair form runs on the four-wave kernel under set_gemm_w4(2))
M, I, K = 32768, 8192, 2048
x, wgu = bf(M, K), bf(2 * I, K, sc=0.05)
t, bl = bf(M, 64), bf(2 * I, 64, sc=0.05)
ab("NT gate|up + GLU", lambda: ops.gemm_nt_glu(x, wgu), 2.0 * M * 2 * I * K)
ops.set_gemm_w4(2)
ab("NT gate|up + GLU + LoRA pair (four-wave)", lambda: ops.gemm_nt_glu(x, wgu, a2=t, b2=bl), 2.0 * M * 2 * I * (K + 64))
ops.set_gemm_w4(False)
t8 = min(timed(lambda: ops.gemm_nt_glu(x, wgu, a2=t, b2=bl)) for _ in range(3))
print(f"NT gate|up + GLU + LoRA pair (eight-wave): {t8:.3f} ms", flush=True)
ops.set_gemm_w4(True)
wd, td, bd = bf(2048, 8192, sc=0.05), bf(M, 64), bf(2048, 64, sc=0.05)
h = bf(M, 8192)
ab("NT down + LoRA pair", lambda: ops.gemm_nt(h, wd, a2=td, b2=bd), 2.0 * M * 2048 * (8192 + 64))
