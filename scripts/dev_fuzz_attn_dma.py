"""Dev-only soak: random shapes / paddings through the attention kernels with LDS-DMA staging against the register-staged kernels (forward and backward, bit for
bit): head_dim 64 (the round-2 DMA kernels), or HEAD_DIM=256 (round 4's three kernels).  Usage: [HEAD_DIM=256] dev_fuzz_attn_dma.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n = time.time() + budget, 0
D = int(os.environ.get("HEAD_DIM", "64"))
scale = D ** -0.5
try:
    while time.time() < t_end:
        Hkv = rng.choice([1, 2, 4, 8]); G = rng.choice([1, 2, 4, 8]); Hq = Hkv * G
        S = rng.choice([rng.randint(1, 130), rng.randint(60, 700), rng.randint(900, 1100), rng.randint(1900, 2100)])
        B = rng.randint(1, max(1, min(32, (1 << 22) // (S * Hq * (D // 64)))))
        g = torch.Generator(device="cuda").manual_seed(rng.randrange(1 << 30))
        qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda", generator=g).to(torch.bfloat16)
        do = torch.randn(B * S, Hq * D, device="cuda", generator=g).to(torch.bfloat16)
        mask = torch.ones(B, S, device="cuda")
        style = rng.random()
        for b in range(B):
            if style < 0.4: mask[b, : rng.randint(0, S - 1)] = 0                      # left padding
            elif style < 0.5: mask[b, rng.randint(0, S - 1):] = 0                     # right padding
            elif style < 0.6: mask[b] = (torch.rand(S, device="cuda", generator=g) > 0.3).float()   # holes
        ops.set_attn_fwd_staging(0)
        o0, l0 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
        d0 = ops.attn_bwd(qkv, mask, o0, do, l0, B, S, Hq, Hkv, D, scale)
        ops.set_attn_fwd_staging(1)
        for rep in range(2):
            o1, l1 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
            assert torch.equal(o0, o1) and torch.equal(l0, l1), ("fwd", B, S, Hq, Hkv, style)
            d1 = ops.attn_bwd(qkv, mask, o0, do, l0, B, S, Hq, Hkv, D, scale)
            assert torch.equal(d0, d1), ("bwd", B, S, Hq, Hkv, style, (d0.float() - d1.float()).abs().max().item())
        n += 1
finally:
    ops.set_attn_fwd_staging(2)
print(f"attention DMA soak ok: {n} random shapes, forward and backward bit for bit in {budget:.0f} s")
