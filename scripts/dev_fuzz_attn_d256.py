"""Dev-only: randomised soak of the head_dim-256 attention kernels (round 6's dQ and dK / dV, round 4's forward) against the register-staged kernels, bit for bit: random batch,
sequence length (1 .. 1500, weighted towards the tile edges), heads per KV head, left padding, several launches per shape.  Usage: python scripts/dev_fuzz_attn_d256.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from ecg_byte_amd import decoder_ops as ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
D, t0, cases = 256, time.time(), 0
edges = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 511, 512, 513]
while time.time() - t0 < budget:
    S = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, 1500))
    Hkv = int(rng.choice([1, 1, 2, 4])); G = int(rng.choice([1, 2, 4, 8])); Hq = Hkv * G
    B = int(rng.integers(1, 4))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    qkv = (torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda", generator=g) * float(rng.choice([0.3, 1.0, 3.0]))).to(torch.bfloat16)
    do = torch.randn(B * S, Hq * D, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.ones(B, S, device="cuda")
    if rng.random() < 0.6:
        for b in range(B): mask[b, : int(rng.integers(0, S + 1))] = 0          # (a row may be all padding)
    ops.set_attn_fwd_staging(0)
    o0, l0 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / 16)
    d0 = ops.attn_bwd(qkv, mask, o0, do, l0, B, S, Hq, Hkv, D, 1 / 16)
    ops.set_attn_fwd_staging(2)
    for rep in range(2):
        o1, l1 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / 16)
        d1 = ops.attn_bwd(qkv, mask, o0, do, l0, B, S, Hq, Hkv, D, 1 / 16)
        if not (torch.equal(o0, o1) and torch.equal(l0.nan_to_num(posinf=1e30), l1.nan_to_num(posinf=1e30)) and torch.equal(d0, d1)):
            print("MISMATCH", dict(B=B, S=S, Hq=Hq, Hkv=Hkv, rep=rep)); raise SystemExit(1)
    cases += 1
print(f"head_dim-256 attention fuzz ok: {cases} shapes in {time.time() - t0:.0f} s")
