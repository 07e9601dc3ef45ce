"""Dev-only: which torch.empty() of a training step is read before it is written?  torch.empty is wrapped: call number i of a step returns memory filled with 0xFF bytes
(NaN as bf16 / fp32, -1 as an integer), every other call zeros; a step whose loss or gradients differ from the all-zeros step names the call site."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
state = {"i": 0, "poison": -1, "sites": {}}

def _fill(t):
    if t.is_cuda and t.numel():
        k = state["i"]; state["i"] += 1
        if k not in state["sites"]:
            fr = traceback.extract_stack(limit=4)[0]
            state["sites"][k] = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90]}"
        flat = t.view(-1) if t.is_contiguous() else None
        if flat is not None:
            flat.view(torch.uint8).fill_(0xFF if (k == state["poison"] or state["poison"] == -2) else 0)
    return t

torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))

def llama_step():
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    cfg = DecoderConfig(vocab_size=5003, hidden_size=512, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, head_dim=64)
    m = HipCausalLM(cfg, seed=5); m.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, 5000, (4, 512), device="cuda", generator=g)
    mask = torch.ones(4, 512, device="cuda"); mask[1, :100] = 0
    labels = torch.full((4, 512), -100, device="cuda"); labels[:, -60:] = ids[:, -60:]
    state["i"] = 0
    out = m(input_ids=ids, attention_mask=mask, labels=labels)
    out.loss.backward()
    return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

def gpt2_step():
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM
    cfg = GPT2Config(vocab_size=4099, n_layer=2)
    m = HipGPT2LM(cfg, seed=5); m.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, 4098, (4, 1024), device="cuda", generator=g)
    mask = torch.ones(4, 1024, device="cuda"); mask[1, :200] = 0; mask[3, :17] = 0
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
    labels = torch.full((4, 1024), -100, device="cuda"); labels[:, -50:] = ids[:, -50:]
    state["i"] = 0
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

for name, fn in (("llama", llama_step), ("gpt2", gpt2_step)):
    state["poison"] = -1; state["sites"] = {}
    l0, g0 = fn()
    n = state["i"]
    state["poison"] = -2
    la, ga = fn()
    print(f"{name}: {n} torch.empty calls a step; all poisoned: loss {float(l0):.6f} -> {float(la):.6f}, gradients that differ {sum(not torch.equal(g0[k], ga[k]) for k in g0)}")
    for i in range(n):
        state["poison"] = i
        li, gi = fn()
        bad = [k for k in g0 if not torch.equal(g0[k], gi[k])]
        if not torch.equal(l0, li) or bad:
            print(f"  call {i}: {state['sites'].get(i)}  loss {float(li):.6f}  gradients that differ: {bad[:5]}")
