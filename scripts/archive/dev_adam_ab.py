"""Dev-only: the multi-tensor Adam launch on the parameter list of Llama-3.2-1B (full fine-tune): ms per step and TB/s over the 22 bytes per parameter it must move.
ECGB_SO=<other build>.so runs another build of the library; two runs in one gpurun call are the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
if os.environ.get("ECGB_SO"): _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])
from ecg_byte_amd import decoder_ops as ops
dev = torch.device("cuda")
H, I, L, V = 2048, 8192, 16, 128256
shapes = []
for _ in range(L):
    shapes += [(3072 * H,), (H * H,), (2 * I * H,), (H * I,), (H,), (H,)]
shapes += [(V * H,), (H,)]
g = torch.Generator(device="cuda").manual_seed(0)
ps = [(torch.randn(s, device=dev, generator=g) * 0.02).bfloat16() for s in shapes]
gs = [(torch.randn(s, device=dev, generator=g) * 0.01).bfloat16() for s in shapes]
ms = [torch.zeros(s, device=dev) for s in shapes]
vs = [torch.zeros(s, device=dev) for s in shapes]
n = sum(p.numel() for p in ps)
plan = ops.SumsqPlan([p.numel() for p in ps], dev)
tabs = ops.AdamMultiPlan(ps, gs, ms, vs, dev)
acc = torch.ones(1, device=dev)
def step(k): ops.adam_multi_(tabs, plan, acc, 1.0, 1e-4, 0.9, 0.99, 1e-8, 1e-2, k)
step(1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(2, 12): step(k)
e1.record(); torch.cuda.synchronize()
ms_ = e0.elapsed_time(e1) / 10
chk = sum(float(p.float().sum()) for p in ps[:6]) + float(ms[2].sum()) + float(vs[3].sum())
print(f"{os.environ.get('ECGB_SO', 'in-tree')}: {n / 1e9:.3f} G parameters, {ms_:.3f} ms per step, {n * 22 / ms_ / 1e9:.2f} TB/s = {n * 22 / ms_ / 1e9 / 8:.3f} of HBM; checksum {chk!r}")
