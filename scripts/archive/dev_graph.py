import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
V = 256000 + 256 + 3500 + 3
cfg = DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1)
m = HipCausalLM(cfg); m.eval()
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.randint(1000, 100000, (1, 600), device="cuda", generator=g); pm = torch.ones_like(p, dtype=torch.float32)
for new in (8, 136, 264):
    for ug in (False, True):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=new, pad_token_id=V - 1, use_graph=ug)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"new {new} graph {ug}: {1e3*dt:.0f} ms")
