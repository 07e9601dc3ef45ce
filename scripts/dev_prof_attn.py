"""Dev-only: phase timers of the forward attention kernel from the -DECGB_PROFILE build (make -C ecg_byte_amd/csrc prof)."""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
from ecg_byte_amd import decoder_ops as ops
B, S, Hq, Hkv, D = [int(v) for v in os.environ.get("SHAPE", "32,1024,32,8,64").split(",")]      # SHAPE=8,2048,8,1,256: the C5 shape
NWAVES = int(os.environ.get("LEAN_WAVES", "8"))
ops.set_attn_lean_waves(NWAVES)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
L = _lib.lib()
L.ecgb_debug_attn_profile.argtypes = [C.c_void_p, C.c_int]
for _ in range(2): ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
torch.cuda.synchronize()
L.ecgb_debug_attn_profile(None, 1)
ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D)); torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
L.ecgb_debug_attn_profile(out, 0)
names = ["issue: DMA of a later tile", "S: K reads + 8 MFMA + V^T reads", "softmax: exp sweep (+ exact path)", "PV: 8 MFMA issue", "wait: vmcnt of the next tile", "barrier"]
for w in range(NWAVES if D == 64 else 4):
    o = out[8 * w: 8 * w + 8]
    trips = o[6]
    tot = sum(o[k] for k in range(6))
    print(f"wave {w}: trips {trips}, cycles per trip {tot / trips:.0f}: " + "  ".join(f"{n.split(':')[0]} {o[k] / trips:.0f}" for k, n in enumerate(names)))
