import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd import decoder_ops as ops
M, N, K = 32768, 16384, 2048
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.gemm_tn(dy, x)
    ops.gemm_nt(x, w)
torch.cuda.synchronize()
