"""Dev-only: the down-projection site's backward of a LoRA fine-tune -- gemm_nn + lora_dx_glu (two launches, round 3) against gemm_nn_glu_bwd_lora (one launch on
the four-wave kernel, round 4): same bits asserted (both activations, with and without dropout, a multi-round and a single-round shape), times on C3's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(torch.bfloat16)
for (M, I, K) in [(32768, 8192, 2048), (8192, 8192, 2048), (16384, 4096, 4096)]:
    dy, w, gu = bf(M, K), bf(K, I, sc=K ** -0.5), bf(M, 2 * I)
    dt, At = bf(M, 64, sc=0.5), bf(I, 64, sc=0.05)
    dt[:, 16:] = 0; At[:, 16:] = 0
    for gelu in (False, True):
        for p, seed in ((0.05, 12345), (0.0, 0), (0.3, (7 << 32) + 99)):
            two = ops.lora_dx_glu(ops.gemm_nn(dy, w), dt, At, gu, 2.0, p, seed, gelu_tanh=gelu)
            one = ops.gemm_nn_glu_bwd_lora(dy, w, gu, dt, At, 2.0, p, seed, gelu_tanh=gelu)
            if one is None:
                print(f"[{M}, {K}] . [{K}, {I}]: the four-wave kernel does not take the shape"); break
            same = torch.equal(one, two)
            nd = (one != two).sum().item()
            print(f"[{M}, {K}] . [{K}, {I}] gelu {gelu} p {p}: same bits {same} ({nd} of {one.numel()} differ)", flush=True)
            assert same
    if M == 32768:
        t1 = min(timed(lambda: ops.gemm_nn_glu_bwd_lora(dy, w, gu, dt, At, 2.0, 0.05, 12345)) for _ in range(3))
        t2a = min(timed(lambda: ops.gemm_nn(dy, w)) for _ in range(3))
        dx = ops.gemm_nn(dy, w)
        t2b = min(timed(lambda: ops.lora_dx_glu(dx, dt, At, gu, 2.0, 0.05, 12345)) for _ in range(3))
        t3 = min(timed(lambda: ops.gemm_nn_glu_bwd(dy, w, gu)) for _ in range(3))
        print(f"one launch {t1:.3f} ms   two launches {t2a:.3f} + {t2b:.3f} = {t2a + t2b:.3f} ms   (product + GLU backward without the adapter: {t3:.3f} ms)", flush=True)
        del dx
    del dy, w, gu, dt, At
