"""Dev-only soak: ecgb_attn_decode_one (RoPE + append + split attention in one launch, workgroups meeting inside the launch) against ecgb_rope_append + ecgb_attn_decode_split
on random shapes, bit for bit (output and cache), host and device lengths, repeated launches on one scratch.  Usage: dev_fuzz_decode_one.py [seconds] [seed]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
t0, n_cases, n_launches = time.time(), 0, 0
while time.time() - t0 < budget:
    D = (64, 128, 256)[ri(0, 2)]
    Hkv = (1, 2, 4, 8)[ri(0, 3)]
    G = (1, 2, 4, 8)[ri(0, 3)]
    Hq = Hkv * G
    B = ri(1, 2)
    cap = 128 * ri(1, 24)
    n = ri(1, cap)
    ns = ri(2, 32)
    if not ops.decode_one_ok(B, Hq, D, ns, cap):
        continue
    g = torch.Generator(device="cuda").manual_seed(ri(0, 1 << 30))
    qkv = (torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda", generator=g) * ri(1, 4)).to(torch.bfloat16)
    cache = torch.randn(B, cap, 2 * Hkv * D, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.ones(B, cap, device="cuda")
    mask[:, n:] = 0
    for b in range(B):
        lead = ri(0, max(0, n - 1))
        if ri(0, 1):
            mask[b, :lead] = 0
        if ri(0, 7) == 0:
            mask[b, :n] = 0                                    # a fully masked row: every score -inf
    pos = torch.rand(B, device="cuda", generator=g) * n
    fr = pos[:, None] * torch.rand(D // 2, device="cuda", generator=g)[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    scale = 1.0 / math.sqrt(D)
    scratch = ops.decode_one_scratch(B, Hq, D, ns, "cuda")
    for dyn in (False, True):
        kv = torch.full((1,), n, dtype=torch.int32, device="cuda") if dyn else n
        q2, c2 = qkv.clone(), cache.clone()
        ops.rope_append_(q2, cos, sin, Hq, Hkv, D, c2, kv)
        ref = ops.attn_decode_split(q2, c2, mask, kv, Hq, Hkv, D, scale, ns)
        for rep in range(2):
            c1 = cache.clone()
            got = ops.attn_decode_one(qkv, cos, sin, c1, mask, kv, Hq, Hkv, D, scale, ns, scratch=scratch)
            n_launches += 1
            same = torch.equal(c1, c2) and (torch.equal(got, ref) or torch.equal(got.view(torch.int16), ref.view(torch.int16)))
            if not same:
                print("MISMATCH", dict(B=B, Hq=Hq, Hkv=Hkv, D=D, cap=cap, n=n, ns=ns, dyn=dyn, rep=rep))
                sys.exit(1)
    n_cases += 1
torch.cuda.synchronize()
print(f"{n_cases} shapes, {n_launches} launches of attn_decode_one_kernel equal rope_append + attn_decode_split bit for bit ({time.time() - t0:.0f} s, seed {seed})")
