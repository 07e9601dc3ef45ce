"""Dev-only: the head_dim-256 forward with LDS-DMA staging against the register-staged kernel (ecgb_set_attn_fwd_staging(0)) -- where do they differ?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = [int(v) for v in os.environ.get("SHAPE", "1,256,8,1,256").split(",")]
torch.manual_seed(0)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
if os.environ.get("PADS"): mask[:, :37] = 0
ops.set_attn_fwd_staging(0)
o0, l0 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
ops.set_attn_fwd_staging(2)
o1, l1 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
print("lse equal", torch.equal(l0.nan_to_num(posinf=1e30), l1.nan_to_num(posinf=1e30)), "o equal", torch.equal(o0, o1), "nan", bool(torch.isnan(o1.float()).any()))
d = (o0.float() - o1.float()).abs().view(B, S, Hq, D)
print("max diff", d.max().item(), "old absmax", o0.float().abs().max().item())
print("by 32-column block:", [round(d[..., i * 32:(i + 1) * 32].max().item(), 4) for i in range(D // 32)])
print("by row block of 32:", [round(d[:, i * 32:(i + 1) * 32].max().item(), 4) for i in range(min(S // 32, 16))])
print("by head:", [round(d[:, :, i].max().item(), 4) for i in range(Hq)])
print("by column within a 32-block (max over blocks):", [round(d.view(B, S, Hq, D // 32, 32)[..., i].max().item(), 3) for i in range(32)])
