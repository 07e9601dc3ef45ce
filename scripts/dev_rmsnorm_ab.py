"""Dev-only: rmsnorm_fwd (+ residual add) at [32768, 2048], the register-resident kernel against the generic one, interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops


def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
x = torch.randn(32768, 2048, device="cuda").to(torch.bfloat16); r = torch.randn(32768, 2048, device="cuda").to(torch.bfloat16); w = torch.randn(2048, device="cuda").to(torch.bfloat16)
res = {0: [], 1: []}
for rnd in range(4):
    for on in (0, 1):
        ops.set_rmsnorm_fwd_rows(bool(on))
        res[on].append(timed(lambda: ops.rmsnorm_fwd(x, w, 1e-6, residual=r)))
ops.set_rmsnorm_fwd_rows(True)
for on in (0, 1):
    t = min(res[on])
    print(f"rows-in-registers {on}: {t * 1e3:.1f} us  {4 * x.numel() * 2 / t / 1e9:.2f} TB/s")

# backward (+ residual gradient), with and without the weight gradient, by rows per workgroup
from ecg_byte_amd import _lib
L = _lib.lib()
dy = torch.randn(32768, 2048, device="cuda").to(torch.bfloat16)
_, rstd, _ = ops.rmsnorm_fwd(x, w, 1e-6)
for with_dw in (True, False):
    res = {}
    for rnd in range(3):
        for n in (64, 32, 16, 8, 4):
            L.ecgb_set_rmsnorm_bwd_rows_per_wg(n)
            L.ecgb_set_rmsnorm_bwd_grid_cap(8192)
            dw = torch.zeros(2048, device="cuda") if with_dw else None
            res.setdefault(n, []).append(timed(lambda: ops.rmsnorm_bwd(x, w, rstd, dy, dw, dres=r)))
    print("rmsnorm_bwd", "with dw" if with_dw else "frozen weights", "  ".join(f"{n} rows/wg: {min(v) * 1e3:.1f} us ({4 * x.numel() * 2 / min(v) / 1e9:.2f} TB/s)" for n, v in res.items()))

# two builds of the library in one process (ECGB_SO_B): backward at the default rows per workgroup
if os.environ.get("ECGB_SO_B"):
    libs = {"A": _lib.lib()}
    _lib._lib = None
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO_B"])
    libs["B"] = _lib.lib()
    for with_dw in (True, False):
        res, outs = {k: [] for k in libs}, {}
        for rnd in range(4):
            for k, L2 in libs.items():
                _lib._lib = L2
                dw = torch.zeros(2048, device="cuda") if with_dw else None
                outs[k] = (ops.rmsnorm_bwd(x, w, rstd, dy, dw, dres=r).clone(), None if dw is None else dw.clone())
                res[k].append(timed(lambda: ops.rmsnorm_bwd(x, w, rstd, dy, dw, dres=r)))
        same = torch.equal(outs["A"][0], outs["B"][0]) and (not with_dw or torch.equal(outs["A"][1], outs["B"][1]))
        print("rmsnorm_bwd", "with dw" if with_dw else "frozen weights", "  ".join(f"{k}: {min(v) * 1e3:.1f} us ({4 * x.numel() * 2 / min(v) / 1e9:.2f} TB/s)" for k, v in res.items()), "  same bits", same)
