"""Dev-only: the head_dim-64 attention backward alone at the C3 shape (for rocprofv3); ECGB_SO picks a build of the library."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
if os.environ.get("ECGB_SO"): _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
o, l = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
for _ in range(20): ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
torch.cuda.synchronize()
