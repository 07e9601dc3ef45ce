"""Dev-only soak: the single-module LoRA sites' input gradient with the adapter share in the GEMM's epilogue (gemm_nn_glu_bwd_lora, gemm_nn_lora) against the
two-kernel paths on random shapes the four-wave kernel takes, random dropout rates and seeds: bit for bit.  Usage: dev_fuzz_lora_fold.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n, skipped = time.time() + budget, 0, 0
bf = lambda g, *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
while time.time() < t_end:
    M, I, K = rng.choice([4096, 8192, 16384, 32768]), rng.choice([2048, 4096, 8192]), rng.choice([1024, 2048, 4096, 8192])
    g = torch.Generator(device="cuda").manual_seed(rng.randrange(1 << 30))
    dy, w = bf(g, M, K), bf(g, K, I, sc=K ** -0.5)
    dt, At = bf(g, M, 64, sc=0.5), bf(g, I, 64, sc=0.05)
    dt[:, 16:] = 0; At[:, 16:] = 0
    p, seed = rng.choice([0.0, 0.05, 0.1, 0.5]), rng.randrange(1 << 40)
    one = ops.gemm_nn_lora(dy, w, dt, At, 2.0, p, seed)
    if one is None: skipped += 1; continue
    assert torch.equal(one, ops.lora_dx_(ops.gemm_nn(dy, w), dt, At, 1, 1, 2.0, p, seed)), ("plain", M, I, K, p, seed)
    if M * I <= (1 << 27):
        gu = bf(g, M, 2 * I)
        for gelu in (False, True):
            a = ops.gemm_nn_glu_bwd_lora(dy, w, gu, dt, At, 2.0, p, seed, gelu_tanh=gelu)
            assert a is not None and torch.equal(a, ops.lora_dx_glu(ops.gemm_nn(dy, w), dt, At, gu, 2.0, p, seed, gelu_tanh=gelu)), ("glu", gelu, M, I, K, p, seed)
    n += 1
print(f"LoRA fold soak ok: {n} shapes bit for bit ({skipped} shapes the four-wave kernel does not take) in {budget:.0f} s")
