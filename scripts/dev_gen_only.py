"""Dev-only: greedy generate at Gemma-2B dims (C5 shape), for timing / rocprofv3 --stats."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
V = 256000 + 256 + 3500 + 3
cfg = DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1)
m = HipCausalLM(cfg)
if _os.environ.get("LORA"):
    m.enable_lora(16, 32, 0.05)
    for n_, p_ in m.named_parameters():
        if "lora_B" in n_ or n_.endswith(".B"):
            p_.data.normal_(0, 0.01)
m.eval()
g = torch.Generator(device="cuda").manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = torch.randint(1000, 100000, (B, 600), device="cuda", generator=g)
pm = torch.ones_like(p, dtype=torch.float32)
for use_graph in ((False, True) if len(sys.argv) > 2 else (False,)):
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=128, pad_token_id=V - 1, use_graph=use_graph)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"B {B} graph {int(use_graph)}: 600-token prompt + 128 new tokens: {1e3*dt:.0f} ms ({B*128/dt:.1f} tokens/s)")
