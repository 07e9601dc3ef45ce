"""Dev-only: time the C5 workload shape -- Gemma-2B dims, seq 2048, LoRA r16 (what the reference's script runs) -- and greedy generate."""
import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT)
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = 2048
V = 256000 + 256 + 3500 + 3
cfg = DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1)
t = time.time(); m = HipCausalLM(cfg); torch.cuda.synchronize(); print("init %.1f s  params %.2f B" % (time.time() - t, sum(p.numel() for p in m.parameters()) / 1e9))
m.enable_lora(r=16, alpha=32, dropout=0.05)
opt = m.make_optimizer()
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(1000, 100000, (B, S), device="cuda", generator=g)
mask = torch.ones(B, S, device="cuda"); mask[:, :100] = 0; ids[:, :100] = cfg.pad_token_id
pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
labels = torch.full((B, S), -100, device="cuda"); labels[:, -20:] = ids[:, -20:]
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.zero_grad()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    opt.step_and_update_lr()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"gemma-2b lora it{it}: {1e3*dt:.0f} ms  {B/dt:.2f} samples/s  loss {out.loss.item():.4f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
m.eval()
p = ids[:1, -600:].contiguous(); pm = torch.ones_like(p, dtype=torch.float32)
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=128, pad_token_id=V - 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"generate 600-token prompt + 128 new tokens: {1e3*dt:.0f} ms ({128/dt:.1f} tokens/s)")
