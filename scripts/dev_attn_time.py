"""Dev-only: the attention kernels at the C3 shape (32 x 1024 tokens, 32 / 8 heads of 64; SHAPE=B,S,Hq,Hkv,D for another), forward and backward apart, HIP events,
interleaved rounds.
Usage: dev_attn_time.py [waves ...]   (lean workgroup widths to compare, default 4)
ECGB_SO_B=<file in ecg_byte_amd/>: a second build of the library loaded beside the shipped one and timed in the same rounds (boxes differ by 5 %: only an
in-process A/B can see a 3 % change); its outputs are compared with the first build's."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
from ecg_byte_amd import decoder_ops as ops

B, S, Hq, Hkv, D = [int(v) for v in os.environ.get("SHAPE", "32,1024,32,8,64").split(",")]      # SHAPE=8,2048,8,1,256: the C5 shape
scale = 1 / math.sqrt(D)
libs = {"A": _lib.lib()}
if os.environ.get("ECGB_SO_B"):
    _lib._lib = None
    _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO_B"])
    libs["B"] = _lib.lib()


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
mask = torch.ones(B, S, device="cuda")
if os.environ.get("PADS"):
    for b in range(B): mask[b, : (37 * b) % (S // 2)] = 0
waves = [int(a) for a in sys.argv[1:]] or [4]
outs = {}
res = {(k, w): ([], []) for k in libs for w in waves}
for rnd in range(5):
    for k, L in libs.items():
        _lib._lib = L
        for w in waves:
            ops.set_attn_lean_waves(w)
            o, l = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
            outs[(k, w)] = (o, l, ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, scale))
            res[(k, w)][0].append(timed(lambda: ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)))
            res[(k, w)][1].append(timed(lambda: ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, scale)))
ff = 4 * B * Hq * S * S * D / 2
for (k, w), (fs, bs) in res.items():
    f, b = min(fs), min(bs)
    print(f"{k} {w} waves: fwd {f:.3f} ms ({ff / f / 1e9:.0f} TF/s)   bwd {b:.3f} ms ({2.5 * ff / b / 1e9:.0f} TF/s)   fwd + bwd {f + b:.3f} ms   (medians {sorted(fs)[2]:.3f} {sorted(bs)[2]:.3f})", flush=True)
if "B" in libs:
    for w in waves:
        a, bb = outs[("A", w)], outs[("B", w)]
        print(f"{w} waves, B against A: o same bits {torch.equal(a[0], bb[0])}  lse {torch.equal(a[1].nan_to_num(posinf=1e30), bb[1].nan_to_num(posinf=1e30))}  d_qkv {torch.equal(a[2], bb[2])}")
