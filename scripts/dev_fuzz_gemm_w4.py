"""Dev-only soak of the four-wave GEMM kernel's round-4 schedule (four barriers per K-tile behind counted waits, buffer_load ... lds pieces): random whole-tile shapes --
1 .. 40 K-tiles, under- and over-filled grids -- through every form it has (NT plain / + LoRA pair / + GLU / + GLU + pair, NN plain / + GLU backward, TN), sent there by
dropping the dispatch threshold to one K-tile per workgroup, against the eight-wave kernels bit for bit, three launches each with other kernels dirtying the memory system in
between.  Usage: dev_fuzz_gemm_w4.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
t_end, n_shapes, n_launch = time.time() + budget, 0, 0
bf = lambda g, *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
while time.time() < t_end:
    tm, tn = rng.randint(1, 48), rng.randint(1, 24)
    if tm * tn > 3000: continue
    M, N = 256 * tm, 256 * tn
    K = 64 * rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 40])
    K2 = rng.choice([0, 64, 128])
    g = torch.Generator(device="cuda").manual_seed(rng.randrange(1 << 30))
    a, b, bn = bf(g, M, K), bf(g, N, K, sc=K ** -0.5), bf(g, K, N, sc=K ** -0.5)
    at = bf(g, K, M)                                                     # TN: a^T . bn, contraction K (multiple of 64)
    kw = dict(a2=bf(g, M, K2), b2=bf(g, N, K2, sc=0.1)) if K2 else {}
    glu = N % 512 == 0
    gb = M * N <= 1 << 26
    gu_in = bf(g, M, 2 * N) if gb else None

    def run():
        out = [ops.gemm_nt(a, b, **kw), ops.gemm_nn(a, bn), ops.gemm_tn(at, bn, splits=1)]
        if glu: out += list(ops.gemm_nt_glu(a, b, **kw))
        if gb: out.append(ops.gemm_nn_glu_bwd(a, bn, gu_in))
        return out
    ops.set_gemm_w4(False)
    want = run()
    ops.set_gemm_w4(True); ops.set_gemm_w4_min_ktiles(1)
    try:
        for rep in range(3):
            noise.random_()
            got = run()
            for k, (x, y) in enumerate(zip(got, want)):
                assert torch.equal(x, y), ("form", k, M, N, K, K2, rep)
            n_launch += len(got)
    finally:
        ops.set_gemm_w4_min_ktiles(128)
    n_shapes += 1
    del a, b, bn, at, kw, gu_in, want, got
print(f"four-wave GEMM soak ok: {n_shapes} shapes, {n_launch} launches compared bit for bit in {budget:.0f} s")
