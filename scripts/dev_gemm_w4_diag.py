"""Dev-only, timing only (wrong results): the four-wave kernel's K-tile schedule with parts removed -- where a K-tile's time goes.
debug mask (sched = 1 + 16 * mask): 1 no barriers, 2 no DMA waits, 4 no DMA at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
REPS = 20
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS
for M, N, K in [(32768, 2048, 8192), (32768, 16384, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    res = {}
    for rnd in range(3):
        for name, s in (("sched 0", 0), ("sched 1", 1), ("no barriers", 17), ("no DMA waits", 33), ("no barriers, no waits", 49), ("no DMA", 65), ("MFMA + LDS reads only", 113)):
            ops.set_gemm_w4_sched(s)
            res.setdefault(name, []).append(timed(lambda: ops.gemm_nt_w4(a, b)))
        res.setdefault("hipBLASLt", []).append(timed(lambda: torch.matmul(a, b.T)))
    ops.set_gemm_w4_sched(1)
    fl = 2.0 * M * N * K
    print(f"[{M}, {K}] -> {N}: " + "   ".join(f"{k} {min(v):.3f} ms ({fl / min(v) / 1e9:.0f} TF/s)" for k, v in res.items()), flush=True)
