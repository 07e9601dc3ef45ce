#!/usr/bin/env python3
"""Regenerates tests/golden/decoder_gemma_tiny.npz from the VENDORED transformers under /root/reference (build container
only): a tiny GemmaForCausalLM (MQA, head_dim 128 != hidden/heads, (1+w) RMSNorm, gelu-tanh gate, sqrt(hidden) embedding
scale, tied embeddings) -- fp32 loss and gradients on a left-padded batch with -100 labels, the bf16 run's loss, and (with every
projection scaled by 4) a greedy `generate` (sequences + per-step fp32 scores + the deviation of the reference's own bf16 run)."""
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
from transformers import GemmaConfig, GemmaForCausalLM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(1)
    cfg = GemmaConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=1, head_dim=128, max_position_embeddings=256, rms_norm_eps=1e-6, rope_theta=10000.0,
                      pad_token_id=299, initializer_range=0.05, hidden_activation="gelu_pytorch_tanh")
    m = GemmaForCausalLM(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layernorm" in n or n.endswith("norm.weight"):
                p.copy_((0.1 * torch.randn_like(p)).to(torch.bfloat16).float())     # Gemma's norm weight is an offset from 1
            else:
                p.copy_(p.to(torch.bfloat16).float())
    B, S = 3, 64
    ids = torch.randint(0, 299, (B, S))
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
            "loss_fp32": np.float32(out.loss.item())}
    for n, p in m.named_parameters():
        data["w:" + n] = p.detach().numpy().copy()     # copies: the parameters are scaled in place below
        data["g:" + n] = p.grad.detach().numpy().copy()
    mb = GemmaForCausalLM(cfg).to(torch.bfloat16).eval()
    mb.load_state_dict({k: v.to(torch.bfloat16) for k, v in m.state_dict().items()})
    data["loss_bf16"] = np.float32(mb(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos).loss.item())
    # greedy generate: prompts of length 45 (not a multiple of 64), left-padded; every projection scaled by 4 (a power of two:
    # still bf16-representable) so that the layers dominate the residual stream and the greedy path wanders
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "proj" in n:
                p.mul_(4.0)
    mb.load_state_dict({k: v.to(torch.bfloat16) for k, v in m.state_dict().items()})
    m.eval()
    g = torch.Generator().manual_seed(11)
    S0, NEW = 45, 16
    pids = torch.randint(0, 299, (B, S0), generator=g)
    pmask = torch.ones(B, S0, dtype=torch.long)
    pmask[0, :9] = 0; pids[0, :9] = 299
    pmask[2, :31] = 0; pids[2, :31] = 299
    gen = m.generate(input_ids=pids, attention_mask=pmask, max_new_tokens=NEW, pad_token_id=299, use_cache=True, do_sample=False,
                     output_scores=True, return_dict_in_generate=True)
    scores = torch.stack(gen.scores, 1)
    full_mask = torch.cat([pmask, torch.ones(B, NEW, dtype=torch.long)], 1)
    fpos = (full_mask.cumsum(-1) - 1).masked_fill(full_mask == 0, 1)
    with torch.no_grad():
        lb = mb(input_ids=gen.sequences, attention_mask=full_mask, position_ids=fpos).logits.float()[:, S0 - 1:-1]
    dev = (lb - scores).abs().amax(-1).numpy()
    data.update(gen_input_ids=pids.numpy(), gen_attention_mask=pmask.numpy(), gen_sequences=gen.sequences.numpy(),
                gen_scores=scores.numpy().astype(np.float32), gen_ref_bf16_deviation=dev.astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "decoder_gemma_tiny.npz"), **data)
    print("loss fp32", out.loss.item(), "bf16", float(data["loss_bf16"]), "generate bf16 deviation max", dev.max())
    print(gen.sequences[:, S0:])


if __name__ == "__main__":
    main()
