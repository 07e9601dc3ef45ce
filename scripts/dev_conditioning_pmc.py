"""Dev-only: the three kernels of the sequence-major conditioning pipeline, three launches each on 4096 records (for rocprofv3: scripts/prof_conditioning.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ecg_byte_amd import preprocess_utils as pp, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = np.ascontiguousarray(synth.synth_ecg(64, 5000, seed=0).transpose(0, 2, 1))
x = np.concatenate([base] * (R // 64)) + 0.01 * np.random.default_rng(0).standard_normal((R, 5000, 12))
xd = torch.from_numpy(x).cuda()
for _ in range(3):
    out, flags, _ = pp._condition_planar(xd, 500, 250, None)
torch.cuda.synchronize()
print("ok", float(out.abs().max()), int(flags.sum()))
