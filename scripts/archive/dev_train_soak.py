"""Dev-only: N optimizer steps of the C3 training step (Llama-3.2-1B dims, B x 1024, random tokens, the same batch every step so
the loss must fall) -- finite losses / gradient norms, steady step time, no memory growth.  Usage: dev_train_soak.py [steps] [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = 1024
V = 128256 + 256 + 3500 + 3
cfg = DecoderConfig.llama_3_2_1b(vocab_size=V, pad_token_id=V - 1)
m = HipCausalLM(cfg)
opt = m.make_optimizer()
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(1000, 100000, (B, S), device="cuda", generator=g)
mask = torch.ones(B, S, device="cuda"); mask[:, :100] = 0; ids[:, :100] = cfg.pad_token_id
pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
labels = torch.full((B, S), -100, device="cuda"); labels[:, -200:] = ids[:, -200:]
losses, times, mem0 = [], [], None
for it in range(steps):
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.zero_grad()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    opt.step_and_update_lr()
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
    losses.append(out.loss.item())
    assert torch.isfinite(out.loss), (it, losses[-5:])
    if it == 5: mem0 = torch.cuda.memory_allocated()
    if it % 10 == 0 or it == steps - 1:
        print(f"step {it:3d}: loss {losses[-1]:.4f}  {1e3*times[-1]:.0f} ms  alloc {torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)
bad = [n for n, p in m.named_parameters() if not torch.isfinite(p.data.float()).all()]
assert not bad, bad
assert torch.cuda.memory_allocated() <= mem0 * 1.01 + (64 << 20), (mem0, torch.cuda.memory_allocated())
print(f"soak ok: {steps} steps, loss {losses[0]:.3f} -> {min(losses):.3f} (last {losses[-1]:.3f}), median step {1e3*sorted(times)[len(times)//2]:.0f} ms, slowest after warm-up {1e3*max(times[5:]):.0f} ms")
