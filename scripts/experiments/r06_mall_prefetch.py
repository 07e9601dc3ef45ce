"""Round 6 experiment: does a read of a weight matrix shortly BEFORE the decode step's few-row product leave it in the Infinity Cache (MALL, 256 MB, memory side) so that
the product -- bound by reading the weights once -- runs faster?  C5 shapes: gate|up [2 x 16384, 2048] (134 MB), down [2048, 16384] (67 MB).
Cases: cold (a 1 GB buffer read in between), warm (the product repeated back to back), prefetched (the weights summed by a plain reduction, then the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ecg_byte_amd import decoder_ops as ops

dev = "cuda"
H, I = 2048, 16384
g = torch.Generator(device=dev).manual_seed(0)
Wgu = (torch.randn(2 * I, H, device=dev, generator=g) * 0.02).bfloat16()
Wd = (torch.randn(H, I, device=dev, generator=g) * 0.02).bfloat16()
x = torch.randn(1, H, device=dev, generator=g).bfloat16()
hm = torch.randn(1, I, device=dev, generator=g).bfloat16()
big = torch.empty(1 << 28, dtype=torch.int32, device=dev).zero_()       # 1 GiB


def timed(fn, before, reps=20):
    ts = []
    for _ in range(reps):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


flush = lambda: big.sum()
for name, fn, W in (("gate|up + GLU (134 MB)", lambda: ops.gemm_nt_glu(x, Wgu, gelu_tanh=True, keep_gu=False), Wgu), ("down (67 MB)", lambda: ops.gemm_nt(hm, Wd), Wd)):
    fn(); torch.cuda.synchronize()
    cold = timed(fn, flush)
    warm = timed(fn, fn)
    pre = timed(fn, lambda: (flush(), W.view(torch.int32).sum()))
    both = timed(fn, lambda: (flush(), Wgu.view(torch.int32).sum(), Wd.view(torch.int32).sum()))
    print(f"{name}: cold {cold:.1f} us, back to back {warm:.1f} us, after a read of the weights {pre:.1f} us, after a read of BOTH matrices (201 MB) {both:.1f} us")
