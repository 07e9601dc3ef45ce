"""Dev-only: is the forward the same bits twice?  (two-layer Llama-3.2-1B dims, B 32, left padding)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_fullshape import LLAMA_1B, LLAMA3_SCALING, _batch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
from ecg_byte_amd import decoder_ops as ops
cfg = DecoderConfig(**{k: v for k, v in LLAMA_1B.items()}, rope_theta=500000.0, rope_scaling=dict(LLAMA3_SCALING), pad_token_id=LLAMA_1B["vocab_size"] - 1)
m = HipCausalLM(cfg, seed=7)
ids, mask, labels, pos = _batch(32, 1024, cfg.vocab_size, cfg.vocab_size - 1, seed=8, pads=[(13 * b) % 700 for b in range(32)], n_labels=33)
def run(full):
    m.full_logits = full
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    return out.loss.item()
print("labelled", [run(False) for _ in range(3)])
print("full    ", [run(True) for _ in range(2)])
fus = ops.glu_fusable
ops.glu_fusable = lambda M, I: False
print("unfused labelled", [run(False) for _ in range(2)], "full", run(True))
ops.glu_fusable = fus
# op-level repeatability
B, S, Hq, Hkv, D = 32, 1024, 32, 8, 64
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
o1, l1 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 0.125)
o2, l2 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 0.125)
print("attn_fwd repeat equal:", torch.equal(o1, o2), torch.equal(l1, l2))
for rep in range(20):
    o2, l2 = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, 0.125)
    if not (torch.equal(o1, o2) and torch.equal(l1, l2)):
        print("attn_fwd differs at repeat", rep); break
else:
    print("attn_fwd: 20 repeats identical")
do = torch.randn(B * S, Hq * D, device="cuda").to(torch.bfloat16)
d1 = ops.attn_bwd(qkv, mask, o1, do, l1, B, S, Hq, Hkv, D, 0.125)
for rep in range(10):
    d2 = ops.attn_bwd(qkv, mask, o1, do, l1, B, S, Hq, Hkv, D, 0.125)
    if not torch.equal(d1, d2):
        print("attn_bwd differs at repeat", rep); break
else:
    print("attn_bwd: 10 repeats identical")
