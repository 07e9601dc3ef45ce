"""Dev-only: one GEMM shape, the eight-wave kernel, the four-wave kernel and hipBLASLt, ten launches each -- for `rocprofv3 --pmc` (scripts/prof_gemm_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
M, N, K = (int(v) for v in os.environ.get("SHAPE", "32768,2048,8192").split(","))
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
SCHEDS = [int(v) for v in os.environ.get("SCHEDS", "0,1").split(",")]      # four-wave schedules (and timing-only diagnostics: 65 no DMA, 113 MFMA + LDS reads only)
for _ in range(10):
    ops.set_gemm_w4(False); ops.gemm_nt(a, b)
    for s in SCHEDS:
        ops.set_gemm_w4_sched(s); ops.gemm_nt_w4(a, b)
    ops.set_gemm_w4_sched(1)
    torch.matmul(a, b.T)
torch.cuda.synchronize()
