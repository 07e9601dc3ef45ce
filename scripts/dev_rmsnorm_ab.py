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
x, r, w = (torch.randn(32768, 2048, device="cuda").to(torch.bfloat16) for _ in range(2)) if False else (None, None, None)
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
