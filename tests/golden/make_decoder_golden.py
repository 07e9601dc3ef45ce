#!/usr/bin/env python3
"""Regenerates tests/golden/decoder_llama_tiny.npz from the VENDORED transformers 4.46.0.dev0 under
/root/reference (build container only).  Tiny Llama config (GQA 2:1, llama3 RoPE scaling, tied
embeddings), a left-padded batch with -100 labels; weights are bf16-representable so the fp32
and bf16 reference runs and the HIP model share them exactly.  Stored: inputs, weights, loss and
gradients of the fp32 run, loss of the bf16 run."""
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
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, rms_norm_eps=1e-5,
                      rope_theta=500000.0, tie_word_embeddings=True, pad_token_id=299, initializer_range=0.05,
                      rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                    "original_max_position_embeddings": 32, "rope_type": "llama3"})
    m = LlamaForCausalLM(cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
        for n, p in m.named_parameters():
            if "layernorm" in n or n.endswith("norm.weight"):
                p.copy_((1.0 + 0.1 * torch.randn_like(p)).to(torch.bfloat16).float())
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
            "loss_fp32": np.float32(out.loss.item()),
            "inv_freq": m.model.rotary_emb.inv_freq.numpy()}
    for n, p in m.named_parameters():
        data["w:" + n] = p.detach().numpy()
        data["g:" + n] = p.grad.detach().numpy()
    mb = LlamaForCausalLM(cfg).to(torch.bfloat16)
    mb.load_state_dict({k: v.to(torch.bfloat16) for k, v in m.state_dict().items()})
    ob = mb(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    data["loss_bf16"] = np.float32(ob.loss.item())
    np.savez_compressed(os.path.join(HERE, "decoder_llama_tiny.npz"), **data)
    print("loss fp32", out.loss.item(), "bf16", ob.loss.item(), "params", sum(p.numel() for p in m.parameters()))


if __name__ == "__main__":
    main()
