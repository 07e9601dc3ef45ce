"""Dev-only: greedy generate at Llama-3.2-1B dims (32 query / 8 KV heads of 64, LoRA r16), 600-token prompt + 128 new tokens, batch 1 and 2: the one-launch decode attention
(32 heads x 12 splits = 384 workgroups a sequence: more than one a CU, all resident) against the four launches -- same tokens asserted, tokens/s of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
V = 128256 + 256 + 4000 + 3
m = HipCausalLM(DecoderConfig.llama_3_2_1b(vocab_size=V, pad_token_id=V - 1))
m.enable_lora(16, 32, 0.05)
for n_, p_ in m.named_parameters():
    if n_.endswith(".B"):
        p_.data.normal_(0, 0.01)
m.eval()
g = torch.Generator(device="cuda").manual_seed(0)
for B in (1, 2):
    p = torch.randint(1000, 100000, (B, 600), device="cuda", generator=g)
    pm = torch.ones_like(p, dtype=torch.float32)
    outs = {}
    for one in (False, True):
        m.decode_attn_one = one
        best = None
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=128, pad_token_id=V - 1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        outs[one] = seq
        print(f"B {B} one-launch attention {one}: {1e3 * best:.0f} ms ({B * 128 / best:.1f} tokens/s)")
    assert torch.equal(outs[False], outs[True])
print("same tokens")
