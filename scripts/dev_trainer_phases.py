"""Dev-only: where a rewrite workgroup's cycles go.  Needs a library whose train.hip was compiled with -DECGB_TRAIN_TIMING (see scripts/README.md), named by ECGB_SO."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ecg_byte_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ.get("ECGB_SO", "libecgbyte_hip_T.so"))
from ecg_byte_amd.trainer import bpe_train_device
if os.environ.get("CORPUS") == "c2":                    # the bench's trainer leg
    import bench
    from ecg_byte_amd.tokenizer import quantize
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import load_tokenizer
    _, _, pc = load_tokenizer("c2")
    x = bench.make_signals(2000, 5000, seed=1, start=0, workers=1)
    text = (quantize(torch.from_numpy(x).cuda(), pc).view(-1) + 97).contiguous()
    del x
else:
    rng = np.random.default_rng(1)
    n = int(os.environ.get("N", 2000 * 12 * 5000))
    steps = rng.integers(-1, 2, size=n, dtype=np.int8)
    sym = np.clip(np.cumsum(steps) % 52, 0, 51); sym = np.where(sym > 25, 51 - sym, sym).astype(np.uint8)
    text = torch.from_numpy(sym + 97).cuda()
L = _lib.lib()
out = (ctypes.c_ulonglong * 16)()
names = ["start: first loads back, arg-max, table init + barrier", "tile: unpack, neighbours' ids, classify", "tile with sites: scan", "-", "tile with sites: walk", "-", "tile without a site: store", "tile with sites: survivors out", "final flush", "neighbours (records, run parity)", "-", "-", "range record", "wait for the workgroup's other waves", "-"]
if os.environ.get("FORM") == "2":
    names = ["start (records, state, table init)", "park (wait for the tile, LDS, barrier)", "span, classify, wave scan", "barrier", "walk", "survivors to registers, barrier", "survivors sent", "flush or barrier", "final flush", "chain of records", "last survivors to registers, barrier", "last survivors sent", "-", "-", "-"]
    from ecg_byte_amd import trainer; trainer.set_train_form(2)
def phases(merges):
    bpe_train_device(text, merges); torch.cuda.synchronize()
    L.ecgb_dev_train_phases(out)
    bpe_train_device(text, merges); torch.cuda.synchronize()
    assert L.ecgb_dev_train_phases(out) == 0
    return [int(out[i]) for i in range(16)]
lo, hi = int(os.environ.get("FROM", "0")), int(os.environ.get("TO", "4000"))
a = phases(lo) if lo > 0 else [0] * 16
b = phases(hi)
d = [y - x for x, y in zip(a, b)]
tot = sum(d[:15])
print(f"merges {lo}..{hi}: waves (form 2: workgroups) {d[15]}, cycles per wave {tot / d[15]:.0f}")
for i, nm in enumerate(names): print(f"  {nm:45s} {100.0 * d[i] / tot:5.1f} %   {d[i] / d[15]:9.0f} cycles per workgroup")
