"""Dev-only: head_dim-64 attention forward and backward, register-staged against LDS-DMA staging: same bits?  time?"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
for B, S, Hq, Hkv, pads in ((32, 1024, 32, 8, True), (32, 1024, 32, 8, False), (3, 1000, 8, 2, True), (2, 2048, 32, 8, True), (5, 70, 4, 4, True)):
    D = 64
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, S, device="cuda")
    if pads:
        for b in range(B): mask[b, : (37 * b) % (S // 2)] = 0
    res = {}
    for dma in (0, 1):
        ops.set_attn_fwd_staging(dma)
        for _ in range(3): o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 1 / math.sqrt(D))
        torch.cuda.synchronize(); res[dma] = ((time.perf_counter() - t) / 10 * 1e3, o, lse)
    do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
    bw = {}
    for dma in (0, 1):
        ops.set_attn_fwd_staging(dma)
        for _ in range(3): d = ops.attn_bwd(qkv, mask, res[0][1], do, res[0][2], B, S, Hq, Hkv, D, 1 / math.sqrt(D))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): d = ops.attn_bwd(qkv, mask, res[0][1], do, res[0][2], B, S, Hq, Hkv, D, 1 / math.sqrt(D))
        torch.cuda.synchronize(); bw[dma] = ((time.perf_counter() - t) / 10 * 1e3, d)
    print(f"   backward: staged {bw[0][0]:.3f} ms  dma {bw[1][0]:.3f} ms  same bits {torch.equal(bw[0][1], bw[1][1])} "
          f"(max diff {(bw[0][1].float() - bw[1][1].float()).abs().max().item():.3g})")
    same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    diff = (res[0][1].float() - res[1][1].float()).abs().max().item()
    print(f"B{B} S{S} {Hq}/{Hkv} pads={pads}: staged {res[0][0]:.3f} ms  dma {res[1][0]:.3f} ms  same bits {same} (max diff {diff:.3g})")
ops.set_attn_fwd_staging(1)
