"""Dev-only soak: random whole-tile shapes through the persistent NT / NN kernels (plain, K-concatenated, GLU epilogue, GLU backward) against the
one-tile-per-workgroup kernels (set_gemm_tile(259)), bit for bit, several launches each while other kernels keep the memory system busy.
Usage: dev_fuzz_gemm_persist.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
t_end, n_shapes, n_launch = time.time() + budget, 0, 0
while time.time() < t_end:
    tm, tn = rng.randint(2, 160), rng.randint(2, 64)
    if tm * tn < 512 or tm * tn > 12000: continue
    M, N = 256 * tm, 256 * tn
    K = 64 * rng.choice([2, 3, 4, 5, 7, 8, 16, 31, 32, 33, 48, 64])
    cat = rng.random() < 0.5
    g = torch.Generator(device="cuda").manual_seed(rng.randrange(1 << 30))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bn = (torch.randn(K, N, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    kw = dict(a2=torch.randn(M, 64, device="cuda", generator=g).to(torch.bfloat16), b2=(torch.randn(N, 64, device="cuda", generator=g) * 0.1).to(torch.bfloat16)) if cat else {}
    glu = ops.glu_fusable(M, N // 2)
    gb = ops.nn_glu_bwd_eligible(M, N, K) and M * N <= 1 << 27
    gu_in = torch.randn(M, 2 * N, device="cuda", generator=g).to(torch.bfloat16) if gb else None
    ops.set_gemm_tile(259)
    want = ops.gemm_nt(a, b, **kw); want_nn = ops.gemm_nn(a, bn)
    want_glu = ops.gemm_nt_glu(a, b, **kw) if glu else None
    want_gb = ops.gemm_nn_glu_bwd(a, bn, gu_in) if gb else None
    ops.set_gemm_tile(0)
    for rep in range(3):
        noise.random_()
        assert torch.equal(ops.gemm_nt(a, b, **kw), want), ("nt", M, N, K, cat, rep)
        assert torch.equal(ops.gemm_nn(a, bn), want_nn), ("nn", M, N, K, rep)
        if glu:
            gu, h = ops.gemm_nt_glu(a, b, **kw)
            assert torch.equal(gu, want_glu[0]) and torch.equal(h, want_glu[1]), ("glu", M, N, K, cat, rep)
        if gb:
            assert torch.equal(ops.gemm_nn_glu_bwd(a, bn, gu_in), want_gb), ("glu_bwd", M, N, K, rep)
        n_launch += 2 + int(glu) + int(gb)
    n_shapes += 1
    del a, b, bn, kw, gu_in, want, want_nn, want_glu, want_gb
print(f"persistent GEMM soak ok: {n_shapes} shapes, {n_launch} launches compared bit for bit in {budget:.0f} s")
