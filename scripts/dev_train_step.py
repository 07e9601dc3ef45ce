"""Dev-only: time the Llama-3.2-1B training step (B x 1024) on HipCausalLM."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = 1024
cfg = DecoderConfig.llama_3_2_1b(vocab_size=128256 + 256 + 3500 + 3, pad_token_id=128256 + 256 + 3500 + 2)
t = time.time(); m = HipCausalLM(cfg); torch.cuda.synchronize(); print("init", time.time() - t, "params", sum(p.numel() for p in m.parameters()) / 1e9)
opt = m.make_optimizer()
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(1000, 100000, (B, S), device="cuda", generator=g)
mask = torch.ones(B, S, device="cuda"); mask[:, :100] = 0; ids[:, :100] = cfg.pad_token_id
pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
labels = torch.full((B, S), -100, device="cuda"); labels[:, -20:] = ids[:, -20:]
modes = [("full-ft", False), ("full-ft+full-logits", True)]
if len(sys.argv) > 2 and sys.argv[2] == "lora":
    m.enable_lora(r=16, alpha=32, dropout=0.05)
    opt = m.make_optimizer()
    modes = [("lora", False)]
for name, full in modes:
    m.full_logits = full
    for it in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        opt.zero_grad()
        out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step_and_update_lr()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"{name} it{it}: fwd {1e3*(t1-t):.0f} ms bwd {1e3*(t2-t1):.0f} ms opt {1e3*(t3-t2):.0f} ms  total {1e3*(t3-t):.0f} ms  {B/(t3-t):.1f} samples/s  loss {out.loss.item():.4f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
