#!/usr/bin/env python3
"""Regenerates tests/golden/decoder_gpt2_tiny.npz from the VENDORED transformers under /root/reference (build container only):
a tiny GPT2LMHeadModel (LayerNorm with bias, learned positions, biased Conv1D c_attn / c_proj / c_fc, gelu_new, tied head, loss
WITHOUT the fp32 upcast of the logits, modeling_gpt2.py:1300-1304) -- fp32 loss and every gradient on a left-padded batch with -100
labels and the reference's position ids, the bf16 run's loss, and a greedy `generate` (sequences + per-step fp32 scores + the
deviation of the reference's own bf16 run).  All dropout probabilities are 0 (parity runs are deterministic)."""
import importlib.metadata as md
import os
import sys

import numpy as np

_orig = md.version


def _fake(name):   # the vendored checkout pins older tokenizers / huggingface-hub (dependency_versions_check.py:57)
    n = name.lower().replace("_", "-")
    return {"tokenizers": "0.20.3", "huggingface-hub": "0.26.0"}.get(n) or _orig(name)


md.version = _fake
sys.path.insert(0, "/root/reference/transformers/src")
import torch  # noqa: E402
from transformers import GPT2Config, GPT2LMHeadModel  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(2)
    cfg = GPT2Config(vocab_size=300, n_positions=128, n_embd=128, n_layer=2, n_head=2, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                     initializer_range=0.05, pad_token_id=299, bos_token_id=297, eos_token_id=298)
    m = GPT2LMHeadModel(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias") and "ln" not in n:
                p.copy_((0.05 * torch.randn_like(p)))                          # the initialiser leaves biases at zero
            if "ln" in n:
                p.copy_(p + 0.1 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())
    B, S = 3, 64
    ids = torch.randint(0, 297, (B, S))
    mask = torch.ones(B, S)
    mask[0, :17] = 0; ids[0, :17] = 299
    mask[2, :40] = 0; ids[2, :40] = 299
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
    pos[mask == 0] = 0
    labels = torch.full((B, S), -100)
    labels[:, -9:] = ids[:, -9:]
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    data = {"input_ids": ids.numpy(), "attention_mask": mask.numpy(), "labels": labels.numpy(), "position_ids": pos.numpy(),
            "loss_fp32": np.float32(out.loss.item()), "logits_fp32": out.logits.detach().numpy()[:, -3:].copy()}
    for n, p in m.named_parameters():
        data["w:" + n] = p.detach().numpy().copy()
        data["g:" + n] = p.grad.detach().numpy().copy()
    mb = GPT2LMHeadModel(cfg).to(torch.bfloat16).eval()
    mb.load_state_dict({k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in m.state_dict().items()})
    data["loss_bf16"] = np.float32(mb(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos).loss.float().item())
    # greedy generate from left-padded prompts of length 45; projections scaled by 4 (a power of two) so that the greedy path wanders
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "c_attn.weight" in n or "c_proj.weight" in n or "c_fc.weight" in n:
                p.mul_(4.0)
    mb.load_state_dict({k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in m.state_dict().items()})
    m.eval()
    g = torch.Generator().manual_seed(11)
    S0, NEW = 45, 16
    pids = torch.randint(0, 297, (B, S0), generator=g)
    pmask = torch.ones(B, S0, dtype=torch.long)
    pmask[0, :9] = 0; pids[0, :9] = 299
    pmask[2, :31] = 0; pids[2, :31] = 299
    gen = m.generate(input_ids=pids, attention_mask=pmask, max_new_tokens=NEW, pad_token_id=299, use_cache=True, do_sample=False,
                     output_scores=True, return_dict_in_generate=True)
    scores = torch.stack(gen.scores, 1)
    full_mask = torch.cat([pmask, torch.ones(B, gen.sequences.shape[1] - S0, dtype=torch.long)], 1)
    fpos = (full_mask.cumsum(-1) - 1).masked_fill(full_mask == 0, 1)
    with torch.no_grad():
        lb = mb(input_ids=gen.sequences, attention_mask=full_mask, position_ids=fpos).logits.float()[:, S0 - 1:-1]
    dev = (lb - scores).abs().amax(-1).numpy()
    data.update(gen_input_ids=pids.numpy(), gen_attention_mask=pmask.numpy(), gen_sequences=gen.sequences.numpy(),
                gen_scores=scores.numpy().astype(np.float32), gen_ref_bf16_deviation=dev.astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "decoder_gpt2_tiny.npz"), **data)
    print("loss fp32", out.loss.item(), "bf16", float(data["loss_bf16"]), "generate bf16 deviation max", dev.max())
    print(gen.sequences[:, S0:])
    print(sorted(k for k in data if k.startswith("w:"))[:40])


if __name__ == "__main__":
    main()
