import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
hidden = 512
cfg = DecoderConfig(vocab_size=5003, hidden_size=hidden, intermediate_size=3 * hidden, num_hidden_layers=2, num_attention_heads=hidden // 64, num_key_value_heads=2, head_dim=64)
m = HipCausalLM(cfg, seed=5); m.train()
g = torch.Generator(device="cuda").manual_seed(3)
ids = torch.randint(0, 5000, (4, 512), device="cuda", generator=g)
mask = torch.ones(4, 512, device="cuda")
labels = torch.full((4, 512), -100, device="cuda"); labels[:, -60:] = ids[:, -60:]
def flat(x, path, out):
    if isinstance(x, torch.Tensor): out.append((path, x.clone()))
    elif isinstance(x, (list, tuple)):
        for i, y in enumerate(x): flat(y, path + (i,), out)
    elif isinstance(x, dict):
        for k, y in x.items(): flat(y, path + (k,), out)
runs = []
for r in range(6):
    out = m(input_ids=ids, attention_mask=mask, labels=labels)
    o = []; flat(m._saved, (), o)
    runs.append((out.loss.detach().clone(), o))
print("losses", [f"{float(l):.6f}" for l, _ in runs])
base = runs[0][1]
for r in range(1, 6):
    diffs = [(p, tuple(t.shape), str(t.dtype)) for (p, t), (_, u) in zip(base, runs[r][1]) if t.shape == u.shape and not torch.equal(t, u)]
    print("run", r, "first tensors that differ:", diffs[:4])
