"""Dev-only: ecgb_ce_fwd_bwd on 4 096 rows x Llama's padded vocabulary (132 608 columns, 132 515 valid), time and HBM rate; ECGB_SO_B: a second build in the same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
from ecg_byte_amd import decoder_ops as ops
libs = {"A": _lib.lib()}
if os.environ.get("ECGB_SO_B"):
    _lib._lib = None
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO_B"])
    libs["B"] = _lib.lib()


def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows, V, ld = 4096, 132515, 132608
torch.manual_seed(0)
src = (torch.randn(rows, ld, device="cuda") * 2).to(torch.bfloat16)
labels = torch.randint(0, V, (rows,), device="cuda"); labels[::7] = -100
inv = torch.tensor([1.0 / 3000], device="cuda")
outs, res = {}, {k: [] for k in libs}
for rnd in range(4):
    for k, L in libs.items():
        _lib._lib = L
        x = src.clone(); lsum = torch.zeros(1, device="cuda")
        rl = ops.ce_fwd_bwd_(x, labels, inv, lsum, V)
        outs[k] = (x, rl)
        y = src.clone()
        res[k].append(timed(lambda: ops.ce_fwd_bwd_(y, labels, inv, lsum, V)))
for k, v in res.items():
    t = min(v)
    print(f"{k}: {t:.3f} ms  {rows * ld * 2 * 2 / t / 1e9:.2f} TB/s ({rows * ld * 4 / t / 1e9 / 8:.3f} of spec)")
if "B" in libs:
    print("same bits: dlogits", torch.equal(outs["A"][0], outs["B"][0]), " row losses", torch.equal(outs["A"][1], outs["B"][1]))
