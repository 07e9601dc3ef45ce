"""Dev-only: where a rewrite workgroup's cycles go.  Needs a library whose train.hip was compiled with -DECGB_TRAIN_TIMING (see scripts/README.md), named by ECGB_SO."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ.get("ECGB_SO", "libecgbyte_hip_T.so"))
from ecg_byte_amd.trainer import bpe_train_device
rng = np.random.default_rng(1)
n = 2000 * 12 * 5000
steps = rng.integers(-1, 2, size=n, dtype=np.int8)
sym = np.clip(np.cumsum(steps) % 52, 0, 51); sym = np.where(sym > 25, 51 - sym, sym).astype(np.uint8)
text = torch.from_numpy(sym + 97).cuda()
L = _lib.lib()
out = (ctypes.c_ulonglong * 16)()
names = ["start (state, table init)", "park (wait for the tile, LDS, barrier)", "span, classify, wave scan", "barrier", "walk", "barrier", "copy out", "flush or barrier", "final flush"]
for merges in (4000,):
    bpe_train_device(text, merges); torch.cuda.synchronize()
    L.ecgb_dev_train_phases(out)
    bpe_train_device(text, merges); torch.cuda.synchronize()
    assert L.ecgb_dev_train_phases(out) == 0
    tot = sum(out[i] for i in range(9))
    print(f"workgroups {out[15]}, cycles per workgroup {tot / out[15]:.0f}")
    for i, nm in enumerate(names): print(f"  {nm:45s} {100.0 * out[i] / tot:5.1f} %   {out[i] / out[15]:9.0f} cycles per workgroup")
