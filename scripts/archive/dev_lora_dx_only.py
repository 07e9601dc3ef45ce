"""Dev-only: the down-projection site's lora_dx + GLU backward pass (and the plain GLU backward beside it) at the step's shape, for the profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
T, I = 32768, 8192
dx = torch.randn(T, I, device="cuda").to(torch.bfloat16)
gu = torch.randn(T, 2 * I, device="cuda").to(torch.bfloat16)
dt = torch.randn(T, 64, device="cuda").to(torch.bfloat16)
At = (torch.randn(I, 64, device="cuda") * 0.02).to(torch.bfloat16)
for _ in range(6):
    d = ops.lora_dx_glu(dx, dt, At, gu, 2.0, 0.05, 1234)
    d2 = ops.glu_bwd(gu, dx)
torch.cuda.synchronize()
