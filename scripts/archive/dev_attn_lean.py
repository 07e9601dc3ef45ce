"""Dev-only: the lean head_dim-64 attention kernels (mode 2) against the round-2 LDS-DMA kernels (mode 1) at the C3 shape: per-kernel time from
HIP events in interleaved rounds (same process, same device), and the error of each against a float64 reference on a small slice."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops

D = 64
scale = 1 / math.sqrt(D)


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B, S, Hq, Hkv, pads in ((32, 1024, 32, 8, False), (32, 1024, 32, 8, True), (4, 1024, 12, 12, False), (8, 2048, 32, 8, False)):
    torch.manual_seed(0)
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, S, device="cuda")
    if pads:
        for b in range(B):
            mask[b, : (37 * b) % (S // 2)] = 0
    ops.set_attn_fwd_staging(1)
    o1, l1 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
    out = {}
    modes = {"r2": (1, 4), "lean": (2, 4), "lean-8waves": (2, 8), "lean-fwd-only": (2 | 0x200 | 0x400, 4), "lean-dq-only": (2 | 0x100 | 0x400, 4), "lean-dkv-only": (2 | 0x100 | 0x200, 4)}
    times = {k: [[], []] for k in modes}
    for rnd in range(4):
        for name, (mode, waves) in modes.items():
            ops.set_attn_fwd_staging(mode)
            ops.set_attn_lean_waves(waves)
            for _ in range(2):
                o, l = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
                d = ops.attn_bwd(qkv, mask, o1, do, l1, B, S, Hq, Hkv, D, scale)
            torch.cuda.synchronize()
            times[name][0].append(timed(lambda: ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)))
            times[name][1].append(timed(lambda: ops.attn_bwd(qkv, mask, o1, do, l1, B, S, Hq, Hkv, D, scale)))
            out[name] = (o, l, d)
    flops_f = 4 * B * Hq * S * S * D / 2
    print(f"== B{B} S{S} {Hq}/{Hkv} pads={pads}")
    for name in modes:
        f, bw = min(times[name][0]), min(times[name][1])
        print(f"   {name:14s} fwd {f:.3f} ms ({flops_f / f / 1e9:.0f} TF/s)  bwd {bw:.3f} ms ({2.5 * flops_f / bw / 1e9:.0f} TF/s)   median fwd {sorted(times[name][0])[2]:.3f} bwd {sorted(times[name][1])[2]:.3f}")
    live = (mask.view(-1) != 0)
    o2, l2, d2 = out["lean"]
    _, _, dr = out["r2"]
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    fin = torch.isfinite(l1)
    print(f"   lean vs r2: o {rel(o2.view(B * S, -1)[live], o1.view(B * S, -1)[live]):.3e}  lse max {((l1[fin] - l2[fin]).abs().max().item()):.3e}  "
          f"d_qkv {rel(d2, dr):.3e}  nan {bool(torch.isnan(o2.float()).any())} {bool(torch.isnan(d2.float()).any())}  "
          f"fin-pattern {torch.equal(fin, torch.isfinite(l2))}")
    # float64 reference on the first batch row
    q, k, v = qkv.view(B, S, Hq + 2 * Hkv, D)[0].double().split([Hq, Hkv, Hkv], dim=1)
    G = Hq // Hkv
    qb, kb, vb = (t.transpose(0, 1).clone().requires_grad_(True) for t in (q, k, v))
    vis = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda")) & (mask[0] != 0)[None, :]
    s = torch.einsum("hqd,hkd->hqk", qb, kb.repeat_interleave(G, 0)) * scale
    p = torch.softmax(s.masked_fill(~vis[None], float("-inf")), -1).nan_to_num(0.0)
    ob = torch.einsum("hqk,hkd->hqd", p, vb.repeat_interleave(G, 0))
    ob.backward(do.view(B, S, Hq, D)[0].double().transpose(0, 1))
    oref = ob.detach().transpose(0, 1).reshape(S, -1)
    dref = torch.cat([qb.grad.transpose(0, 1), kb.grad.transpose(0, 1), vb.grad.transpose(0, 1)], 1).reshape(S, -1)
    lv = live[:S]
    for name in ("r2", "lean"):
        o, l, d = out[name]
        print(f"   {name:5s} vs float64 (row 0): o {rel(o.view(B * S, -1)[:S][lv], oref[lv]):.3e}  d_qkv {rel(d[:S], dref):.3e}")
ops.set_attn_fwd_staging(2)
ops.set_attn_lean_waves(4)
