"""Dev-only: phase timers of encode_flow_kernel from the -DECGB_PROFILE build (make -C ecg_byte_amd/csrc prof): clock64 ticks of every wave's lane 0, summed per workgroup,
for the bench batch (4 096 records of 12 x 5000, C2 tokenizer).  Phases: 0 stage + run map, 1 parse, 6 / 7 / 2 resolve (lengths, chain, prefix counts), 3 emit."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libecgbyte_hip_prof.so")
import bench
from helpers import load_tokenizer
from ecg_byte_amd.tokenizer import HipTokenizer
vocab, merges, pc = load_tokenizer("c2")
B = int(os.environ.get("B", "4096"))
x = torch.from_numpy(bench.make_signals(B, 5000, seed=0, start=0, workers=int(os.environ.get("WORKERS", "8")))).cuda()
tok = HipTokenizer(merges)
L = _lib.lib()
L.ecgb_debug_set_profile_buffer.argtypes = [ctypes.c_void_p]
L.ecgb_debug_set_profile_buffer.restype = None
ids, counts = tok.quantize_encode(x, pc, ids_stride=8192)
torch.cuda.synchronize()
prof = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
L.ecgb_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
ids, counts = tok.quantize_encode(x, pc, ids_stride=8192)
torch.cuda.synchronize()
L.ecgb_debug_set_profile_buffer(ctypes.c_void_p(0))
p = prof.view(-1, 8).sum(0).cpu().numpy().astype(np.float64)
names = {0: "stage + run map", 1: "parse", 6: "resolve: lengths", 7: "resolve: chain", 2: "resolve: prefix counts", 3: "emit"}
tot = sum(p[k] for k in names)
print(f"tokens {int(counts.sum())}, ticks per record {tot / B:.0f}")
for k, nm in names.items(): print(f"  {nm:26s} {100 * p[k] / tot:5.1f} %")
print("  counters [4], [5]:", p[4], p[5])
