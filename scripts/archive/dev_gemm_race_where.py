import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ecg_byte_amd import decoder_ops as ops
torch.manual_seed(0)
M, N, K = 2048, 3072, 512
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
ref = x.float() @ w.float().t()
good = ref.bfloat16()
# 1. torch's own GEMM, same harness
outs = [(x @ w.t()).clone() for _ in range(40)]
print("torch matmul: launches that differ", sum(not torch.equal(outs[0], o) for o in outs[1:]))
# 2. ours with a synchronize between launches
outs = []
for _ in range(40):
    o = ops.gemm_nt(x, w); torch.cuda.synchronize(); outs.append(o.clone())
print("ours, synchronised: launches that differ", sum(not torch.equal(outs[0], o) for o in outs[1:]))
# 3. where are the wrong elements
for o in outs[:6]:
    bad = ((o.float() - ref).abs() > 0.02).nonzero()
    print("elements off by > 0.02:", bad.shape[0], bad[:8].tolist())
# 4. output into a fixed buffer
buf = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
outs = []
for _ in range(40):
    ops.gemm_nt(x, w, out=buf) if "out" in ops.gemm_nt.__code__.co_varnames else None
    outs.append(buf.clone())
print("ours, fixed output buffer: launches that differ", sum(not torch.equal(outs[0], o) for o in outs[1:]))
