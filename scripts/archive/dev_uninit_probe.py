"""Dev-only: does a training step read memory it did not write?  The caching allocator's free blocks are filled with NaN patterns (bf16 / fp32 / fp64 NaN bits, 0xFF bytes)
before every step of a fresh model; the loss and every gradient must stay finite and equal to the first run's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def poison(mb=3000):
    junk = [torch.full((mb * 1024 * 1024 // 4 // 8,), float("nan"), device="cuda") for _ in range(8)]
    for j in junk: j.view(torch.int32).fill_(-1)          # 0xFFFFFFFF: NaN as bf16 pairs, as fp32, as fp64 halves; -1 as an integer
    torch.cuda.synchronize()
    del junk

def gpt2_step():
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM
    cfg = GPT2Config(vocab_size=4099, n_layer=2)
    m = HipGPT2LM(cfg, seed=5); m.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, 4098, (4, 1024), device="cuda", generator=g)
    mask = torch.ones(4, 1024, device="cuda"); mask[1, :200] = 0; mask[3, :17] = 0
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
    labels = torch.full((4, 1024), -100, device="cuda"); labels[:, -50:] = ids[:, -50:]
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

def llama_step(lora):
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    cfg = DecoderConfig(vocab_size=5003, hidden_size=512, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, head_dim=64)
    m = HipCausalLM(cfg, seed=5)
    if lora: m.add_lora(r=16, alpha=32, dropout=0.05, seed=7)
    m.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, 5000, (4, 512), device="cuda", generator=g)
    mask = torch.ones(4, 512, device="cuda"); mask[1, :100] = 0
    labels = torch.full((4, 512), -100, device="cuda"); labels[:, -60:] = ids[:, -60:]
    out = m(input_ids=ids, attention_mask=mask, labels=labels)
    out.loss.backward()
    return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

for name, fn in (("gpt2", gpt2_step), ("llama", lambda: llama_step(False)), ("llama lora", lambda: llama_step(True))):
    try:
        l1, g1 = fn()
    except Exception as e:
        print(name, "could not run:", repr(e)[:200]); continue
    poison()
    l2, g2 = fn()
    bad = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    nonfin = [k for k in g2 if not torch.isfinite(g2[k].float()).all()]
    print(f"{name}: loss {float(l1):.6f} / {float(l2):.6f} same bits {torch.equal(l1, l2)}; gradients that differ: {bad[:6]}{' ...' if len(bad) > 6 else ''}; not finite: {nonfin[:6]}")
