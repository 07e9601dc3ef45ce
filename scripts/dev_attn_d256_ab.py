"""Dev-only: the head_dim-256 attention kernels of several builds of the library in one process (the C5 shape: 8 x 2048 tokens, 8 / 1 heads of 256; SHAPE=B,S,Hq,Hkv,D
for another): forward and backward timed apart with HIP events, interleaved rounds, and the outputs of every build compared bit for bit with the first one's.
Usage: python scripts/dev_attn_d256_ab.py libecgbyte_hip.so libecgbyte_old.so ...   (files in ecg_byte_amd/; PADS=1: left-padded rows)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import _lib
from ecg_byte_amd import decoder_ops as ops

B, S, Hq, Hkv, D = [int(v) for v in os.environ.get("SHAPE", "8,2048,8,1,256").split(",")]
scale = 1 / math.sqrt(D)
names = sys.argv[1:] or ["libecgbyte_hip.so"]
libs = {}
base = os.path.dirname(_lib.SO_PATH)
for n in names:
    _lib._lib = None
    _lib.SO_PATH = os.path.join(base, n)
    libs[n] = _lib.lib()


def timed(fn, reps=10):
    for _ in range(2): fn()
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
outs, res = {}, {n: ([], []) for n in names}
if os.environ.get("PASS_P") is not None:                                   # PASS_P=1: the dK pass hands its probabilities to a dV kernel (default 0: the pair kernel)
    for L in libs.values():
        if hasattr(L, "ecgb_set_attn_d256_pass_p"): L.ecgb_set_attn_d256_pass_p(int(os.environ["PASS_P"]))
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for n, L in libs.items():
        _lib._lib = L
        o, l = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
        outs[n] = (o, l, ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, scale))
        res[n][0].append(timed(lambda: ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)))
        res[n][1].append(timed(lambda: ops.attn_bwd(qkv, mask, o, do, l, B, S, Hq, Hkv, D, scale)))
ff = 4 * B * Hq * S * S * D / 2
for n, (fs, bs) in res.items():
    f, b = min(fs), min(bs)
    print(f"{n:24s} fwd {f:.3f} ms ({ff / f / 1e9:.0f} TF/s)   bwd {b:.3f} ms ({2.5 * ff / b / 1e9:.0f} TF/s at five products)", flush=True)
a = outs[names[0]]
for n in names[1:]:
    bb = outs[n]
    dq_same = torch.equal(a[2][:, : Hq * D], bb[2][:, : Hq * D])
    print(f"{n} against {names[0]}: o same bits {torch.equal(a[0], bb[0])}  lse {torch.equal(a[1].nan_to_num(posinf=1e30), bb[1].nan_to_num(posinf=1e30))}  "
          f"dq {dq_same}  d_qkv {torch.equal(a[2], bb[2])}")
