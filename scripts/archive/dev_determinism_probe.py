import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
def step(pad, pos_given, hidden=512):
    cfg = DecoderConfig(vocab_size=5003, hidden_size=hidden, intermediate_size=3 * hidden, num_hidden_layers=2, num_attention_heads=hidden // 64, num_key_value_heads=2 if hidden >= 128 else 1, head_dim=64)
    m = HipCausalLM(cfg, seed=5); m.train()
    g = torch.Generator(device="cuda").manual_seed(3)
    ids = torch.randint(0, 5000, (4, 512), device="cuda", generator=g)
    mask = torch.ones(4, 512, device="cuda")
    if pad: mask[1, :100] = 0
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long() if pos_given else None
    labels = torch.full((4, 512), -100, device="cuda"); labels[:, -60:] = ids[:, -60:]
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    return out.loss.detach().clone(), m.embed.grad.clone(), m.wqkv[0].grad.clone()
for pad in (False, True):
    for pos_given in (False, True):
        r = [step(pad, pos_given) for _ in range(6)]
        print("pad", pad, "positions given", pos_given, "losses", sorted(set(f"{float(x[0]):.6f}" for x in r)), "embed grad same", all(torch.equal(r[0][1], x[1]) for x in r), "wqkv grad same", all(torch.equal(r[0][2], x[2]) for x in r))
