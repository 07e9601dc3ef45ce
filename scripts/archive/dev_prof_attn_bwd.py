"""Dev-only: phase timers of attn_bwd_dkv_lean_kernel from the -DECGB_PROFILE build (make -C ecg_byte_amd/csrc prof); the forward's are in dev_prof_attn.py."""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
scale = 1 / math.sqrt(D)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16); do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
L = _lib.lib()
L.ecgb_debug_attn_profile.argtypes = [C.c_void_p, C.c_int]
o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
for _ in range(2): ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale)
torch.cuda.synchronize()
L.ecgb_debug_attn_profile(None, 1)
ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale); torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
L.ecgb_debug_attn_profile(out, 0)
names = ["issue (5 DMA pieces)", "stats + first fragments (one LDS round trip) x2", "S / dP: 16 MFMA + second batch + transposed reads x2", "exp, ds, dV / dK: 16 MFMA x2", "wait (vmcnt)", "barrier"]
for w in range(4):
    o_ = out[8 * w: 8 * w + 8]
    trips = max(1, o_[6])
    tot = sum(o_[k] for k in range(6))
    print(f"dK/dV wave {w}: steps {trips}, cycles per step {tot / trips:.0f}: " + "  ".join(f"[{n}] {o_[k] / trips:.0f}" for k, n in enumerate(names)))
