"""Dev-only: phase timers of the head_dim-256 dQ kernel from a -DECGB_PROFILE build of attention.hip (libecgbyte_prof256.so: see EXPERIMENTS.md R6), the C5 shape."""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ.get("SO", "libecgbyte_prof256.so"))
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = [int(v) for v in os.environ.get("SHAPE", "8,2048,8,1,256").split(",")]
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
L = _lib.lib()
L.ecgb_debug_attn256_profile.argtypes = [C.c_void_p, C.c_int]
sc = 1 / math.sqrt(D)
o, l = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, sc)
for _ in range(2): ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, sc)
torch.cuda.synchronize()
L.ecgb_debug_attn256_profile(None, 1)
ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, sc); torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
L.ecgb_debug_attn256_profile(out, 0)
names = ["score groups (+ DMA issue)", "softmax", "accumulate groups", "skipped tile / advance", "vmcnt wait", "barrier"]
# the dQ kernel and the dK pass of the pair kernel both add to slots 0 .. 31 (the dV pass to 32 .. 63): KERNEL=dq profiles a dQ-only launch is not possible through the
# C ABI, so the dQ figures are those of round 6's first profile (EXPERIMENTS.md); here the pair kernel's two passes are told apart, dQ's cycles are in the dK rows too
for label, base in (("dQ + dK pass", 0), ("dV pass", 32)):
    for w in range(4):
        v = out[base + 8 * w: base + 8 * w + 8]
        n = max(1, v[6])
        print(f"{label} wave {w}: {v[6]} active steps; per active step: " + "  ".join(f"{nm} {v[k] / n:.0f}" for k, nm in enumerate(names)) + f"   sum {sum(v[:6]) / n:.0f}")
