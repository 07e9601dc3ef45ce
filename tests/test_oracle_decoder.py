"""CPU test pinning the decoder oracle (oracle/llama_ref.py) to the vendored transformers goldens."""
import os

import numpy as np
import torch

from helpers import GOLDEN
from oracle import llama_ref as R

CFG = dict(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5)
SCALING = {"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 32}


def test_decoder_oracle_matches_vendored_transformers_fp32():
    z = np.load(os.path.join(GOLDEN, "decoder_llama_tiny.npz"))
    inv = R.llama3_inv_freq(64, 500000.0, SCALING)
    assert np.allclose(inv.numpy(), z["inv_freq"], rtol=1e-6, atol=0)
    params = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("w:")}
    loss = R.llama_loss(params, CFG, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                        torch.from_numpy(z["labels"]), torch.from_numpy(z["position_ids"]), inv)
    assert abs(loss.item() - float(z["loss_fp32"])) < 1e-5
    loss.backward()
    for k, p in params.items():
        ref = torch.from_numpy(z["g:" + k])
        assert torch.allclose(p.grad, ref, atol=1e-6, rtol=1e-4), k


def test_decoder_oracle_bf16_close_to_reference_bf16():
    z = np.load(os.path.join(GOLDEN, "decoder_llama_tiny.npz"))
    inv = R.llama3_inv_freq(64, 500000.0, SCALING)
    params = {k[2:]: torch.from_numpy(z[k]).to(torch.bfloat16) for k in z.files if k.startswith("w:")}
    loss = R.llama_loss(params, CFG, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                        torch.from_numpy(z["labels"]), torch.from_numpy(z["position_ids"]), inv)
    assert abs(loss.item() - float(z["loss_bf16"])) < 2e-3


def test_decoder_oracle_gemma_matches_vendored_transformers_fp32():
    """The Gemma branch of the oracle (fp32 (1 + w) RMSNorm, gelu-tanh gate, sqrt(hidden) embedding scale, head_dim 128 !=
    hidden / heads, MQA) against loss and every gradient of the vendored GemmaForCausalLM (make_decoder_golden_gemma.py)."""
    z = np.load(os.path.join(GOLDEN, "decoder_gemma_tiny.npz"))
    cfg = dict(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-6, model_type="gemma")
    inv = R.llama3_inv_freq(128, 10000.0, None)
    params = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("w:") and "lm_head" not in k}
    loss = R.llama_loss(params, cfg, torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                        torch.from_numpy(z["labels"]), torch.from_numpy(z["position_ids"]), inv)
    assert abs(loss.item() - float(z["loss_fp32"])) < 1e-5
    loss.backward()
    for k, p in params.items():
        ref = torch.from_numpy(z["g:" + k])
        assert torch.allclose(p.grad, ref, atol=2e-6, rtol=1e-4), k


def test_gpt2_oracle_matches_vendored_transformers_fp32():
    """oracle/gpt2_ref.py (BASELINE config C1's model) against loss, logits and every gradient of the vendored GPT2LMHeadModel
    (tests/golden/make_decoder_golden_gpt2.py): LayerNorm, learned positions, biased Conv1D projections, gelu_new, tied head."""
    from oracle import gpt2_ref as G
    z = np.load(os.path.join(GOLDEN, "decoder_gpt2_tiny.npz"))
    cfg = dict(vocab_size=300, n_positions=128, n_embd=128, n_layer=2, n_head=2)
    params = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("w:")}
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    labels, pos = torch.from_numpy(z["labels"]), torch.from_numpy(z["position_ids"])
    logits = G.gpt2_logits(params, cfg, ids, mask, pos)
    assert torch.allclose(logits[:, -3:], torch.from_numpy(z["logits_fp32"]), atol=2e-5, rtol=1e-4)
    loss = G.gpt2_loss(params, cfg, ids, mask, labels, pos)
    assert abs(loss.item() - float(z["loss_fp32"])) < 1e-5
    loss.backward()
    for k, p in params.items():
        assert torch.allclose(p.grad, torch.from_numpy(z["g:" + k]), atol=2e-6, rtol=1e-4), k
    pb = {k: v.detach().to(torch.bfloat16) for k, v in params.items()}
    lb = G.gpt2_loss(pb, cfg, ids, mask, labels, pos)                       # the bf16 run: loss computed on bf16 logits, no upcast
    assert abs(lb.float().item() - float(z["loss_bf16"])) < 4e-2
