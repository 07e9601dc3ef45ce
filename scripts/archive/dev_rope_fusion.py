"""Dev-only: attention backward + RoPE backward at the C3 shape, fused (ecgb_attn_bwd_rope) against the two steps apart, interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D, scale = 32, 1024, 32, 8, 64, 0.125
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16); do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
pos = torch.arange(S, device="cuda").repeat(B).float()
fr = pos[:, None] * (1.0 / (500000.0 ** (torch.arange(0, D, 2, device="cuda").float() / D)))[None, :]
cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {True: [], False: []}
for rnd in range(4):
    for on in (False, True):
        ops.set_attn_bwd_rope_fusion(on)
        res[on].append(timed(lambda: ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale, rope=(cos, sin))))
ops.set_attn_bwd_rope_fusion(True)
print(f"attention backward + RoPE backward per layer: apart {min(res[False]):.3f} ms, fused {min(res[True]):.3f} ms")

# forward: the q|k|v projection + RoPE, apart (eight-wave GEMM, then ecgb_rope) against ecgb_gemm_nt_bf16_rope (four-wave kernel, rotation in the epilogue)
H = 2048
x = torch.randn(B * S, H, device="cuda").to(torch.bfloat16); wqkv = (torch.randn((Hq + 2 * Hkv) * D, H, device="cuda") * 0.05).to(torch.bfloat16)
t = torch.randn(B * S, 64, device="cuda").to(torch.bfloat16); bl = (torch.randn((Hq + 2 * Hkv) * D, 64, device="cuda") * 0.05).to(torch.bfloat16)
for name, kw in (("q|k|v projection + RoPE", {}), ("the same with a LoRA pair", dict(a2=t, b2=bl))):
    res = {True: [], False: []}
    for rnd in range(4):
        for on in (False, True):
            ops.set_gemm_rope_fusion(on)
            res[on].append(timed(lambda: ops.gemm_nt_rope(x, wqkv, cos, sin, (Hq + Hkv) * D, **kw)))
    ops.set_gemm_rope_fusion(False)
    print(f"{name}: apart {min(res[False]):.3f} ms, fused {min(res[True]):.3f} ms")
