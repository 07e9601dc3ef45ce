"""Dev-only: phase timers of encode_long_kernel from the -DECGB_PROFILE build (make -C ecg_byte_amd/csrc prof): ticks of every wave's lane 0, summed per workgroup,
for the bench batch.  Phases: 0 quantise + run-length entries, 1 walk of the own chunks (B1), 2 run-on (B2), 3 assembly + copy; counters 4 / 5: loop trips of B1 / B2."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
from helpers import load_tokenizer
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan
vocab, merges, pc = load_tokenizer("c2")
B = int(os.environ.get("B", "4096"))
x = torch.from_numpy(synth.synth_ecg(B, 5000, seed=0)).cuda()
tok = HipTokenizer(merges)
L = _lib.lib()
L.ecgb_debug_set_profile_buffer.argtypes = [ctypes.c_void_p]
L.ecgb_debug_set_profile_buffer.restype = None
set_encode_plan(int(os.environ.get("PLAN", "4")))
ids, counts = tok.quantize_encode(x, pc, ids_stride=8192)
torch.cuda.synchronize()
prof = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
L.ecgb_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
ids, counts = tok.quantize_encode(x, pc, ids_stride=8192)
torch.cuda.synchronize()
L.ecgb_debug_set_profile_buffer(ctypes.c_void_p(0))
p = prof.view(-1, 8).sum(0).cpu().numpy().astype(np.float64)
names = {0: "quantise + run-length entries", 1: "B1: own chunks", 2: "B2: run-on", 3: "assembly + copy"}
tot = sum(p[k] for k in names)
print(f"tokens {int(counts.sum())}, ticks per record {tot / B:.0f}")
for k, nm in names.items(): print(f"  {nm:32s} {100 * p[k] / tot:5.1f} %   {p[k] / B:9.0f} ticks per record")
print(f"  trips per record: B1 {p[4] / B:.0f}, B2 {p[5] / B:.0f};  ticks per B1 trip {p[1] / max(p[4], 1):.0f}")
