"""Dev-only: phase timers of encode_kernel from the -DECGB_PROFILE build."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, os, ctypes as C
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
_, merges, pc = load_tokenizer("c2")
tk = HipTokenizer(merges)
base = synth.synth_ecg(256, 5000, seed=0)
x = np.concatenate([base] * (B // 256))
xd = torch.from_numpy(x).cuda()
nwg = 256
prof = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
L = _lib.lib()
tk.quantize_encode(xd, pc); torch.cuda.synchronize()
L.ecgb_debug_set_profile_buffer(C.c_void_p(prof.data_ptr()))
tk.quantize_encode(xd, pc); torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(nwg, 8)
names = ["stage", "parse", "resolve_passes", "emit", "passes", "trips", "resolve_sweep", "resolve_first_follow"]
for k, nm in enumerate(names):
    print(f"{nm:12s} mean {p[:,k].mean():12.0f}  min {p[:,k].min():10d}  max {p[:,k].max():10d}")
