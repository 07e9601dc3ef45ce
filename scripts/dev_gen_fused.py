"""Dev: greedy generate at Gemma-2B dims (C5: 600-token prompt + 128 new tokens, batch 1, replayed graph) with the fused decode step on and off.
    LORA=1 python scripts/dev_gen_fused.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
V = 256000 + 256 + 3500 + 3
cfg = DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1)
m = HipCausalLM(cfg)
if os.environ.get("LORA", "1") == "1":
    m.enable_lora(16, 32, 0.05)
    for n_, p_ in m.named_parameters():
        if n_.endswith(".B"):
            p_.data.normal_(0, 0.01)
m.eval()
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.randint(1000, 100000, (1, 600), device="cuda", generator=g)
pm = torch.ones_like(p, dtype=torch.float32)
res = {}
for fused in (False, True, 2):
    m.decode_fused, m.decode_fused_attn = bool(fused), fused == 2
    m.__dict__.pop("_gen_graphs", None)
    best = {}
    for n_new in (8, 128):
        b = None
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=n_new, pad_token_id=V - 1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            b = dt if b is None else min(b, dt)
        best[n_new] = b
    per = (best[128] - best[8]) / 120
    res[fused] = seq
    print(f"fused {int(fused)}: 128 new tokens in {1e3 * best[128]:.1f} ms ({128 / best[128]:.1f} tokens/s); decode {1e3 * per:.3f} ms per token = {1 / per:.1f} tokens/s", flush=True)
print("same tokens:", bool(torch.equal(res[True], res[False])), bool(torch.equal(res[2], res[False])))
