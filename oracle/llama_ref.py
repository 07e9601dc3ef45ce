"""Plain-PyTorch restatement of the reference's decoder step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE
(same rules as oracle/oracle.py; also bench.py's cpu_baseline for the train metric).

Follows the vendored transformers 4.46.0.dev0 under /root/reference/transformers/src/transformers:
  LlamaRMSNorm            models/llama/modeling_llama.py:67-72   (normalise in fp32, cast, THEN * weight)
  rotate_half / RoPE      models/llama/modeling_llama.py:193-224, 145-165 (cos/sin cast to the activation dtype)
  LlamaSdpaAttention      models/llama/modeling_llama.py:526-614 with the 4-D causal + padding mask of 1047-1100
  LlamaMLP                models/llama/modeling_llama.py:238-258
  LlamaDecoderLayer       models/llama/modeling_llama.py:635-701
  ForCausalLMLoss         loss/loss_utils.py:24-47 (float upcast, shift, ignore_index -100, mean)
Gemma (cfg["model_type"] == "gemma", config C5's family), models/gemma/modeling_gemma.py:
  GemmaRMSNorm            51-68   (x_hat * (1 + w) in fp32, THEN cast)
  GemmaMLP                131-152 (gelu-tanh gate)
  embeddings * sqrt(hidden) held in the activation dtype, 800-801; attention as Llama's with head_dim from the config
Pinned against outputs of the reference itself: tests/golden/decoder_llama_tiny.npz and decoder_gemma_tiny.npz
(tests/test_oracle_decoder.py, fp32, 1e-5).
"""
import math

import torch
import torch.nn.functional as F


def rms_norm(x, w, eps, gemma=False):
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    if gemma:
        return (xf * (1.0 + w.float())).to(dt)
    return w * xf.to(dt)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _lin(params, name, x, lora_scale):
    """nn.Linear, plus the LoRA branch y += scale * B(A x) when `<name>.lora_A/B` are present (peft LoraLayer, dropout 0)."""
    y = F.linear(x, params[name + ".weight"])
    if lora_scale is not None and (name + ".lora_A") in params:
        y = y + lora_scale * F.linear(F.linear(x, params[name + ".lora_A"]), params[name + ".lora_B"])
    return y


def llama_loss(params, cfg, input_ids, attention_mask, labels, position_ids, inv_freq, lora_scale=None):
    """params: dict of HF-named tensors (requires_grad as wanted).  Returns the scalar loss."""
    logits = llama_logits(params, cfg, input_ids, attention_mask, position_ids, inv_freq, lora_scale)
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=-100)


def llama_logits(params, cfg, input_ids, attention_mask, position_ids, inv_freq, lora_scale=None):
    """Float logits [B, S, V] of LlamaForCausalLM.forward (modeling_llama.py:1135-1213)."""
    H, D = cfg["hidden_size"], cfg["head_dim"]
    Hq, Hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    B, S = input_ids.shape
    emb = params["model.embed_tokens.weight"]
    dt = emb.dtype
    gemma = cfg.get("model_type", "llama") == "gemma"
    x = emb[input_ids]
    if gemma:
        x = x * torch.tensor(H ** 0.5, dtype=dt)
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :].float()
    e = torch.cat((freqs, freqs), -1)
    cos, sin = e.cos().to(dt)[:, None], e.sin().to(dt)[:, None]
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool, device=x.device))
    visible = causal[None, None] & (attention_mask[:, None, None, :] != 0)
    bias = torch.zeros(B, 1, S, S, dtype=dt, device=x.device).masked_fill(~visible, torch.finfo(dt).min)
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        h = rms_norm(x, params[p + "input_layernorm.weight"], cfg["rms_norm_eps"], gemma)
        q = _lin(params, p + "self_attn.q_proj", h, lora_scale).view(B, S, Hq, D).transpose(1, 2)
        k = _lin(params, p + "self_attn.k_proj", h, lora_scale).view(B, S, Hkv, D).transpose(1, 2)
        v = _lin(params, p + "self_attn.v_proj", h, lora_scale).view(B, S, Hkv, D).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        k = k.repeat_interleave(Hq // Hkv, 1)
        v = v.repeat_interleave(Hq // Hkv, 1)
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
        a = a.transpose(1, 2).reshape(B, S, Hq * D)
        x = x + _lin(params, p + "self_attn.o_proj", a, lora_scale)
        h = rms_norm(x, params[p + "post_attention_layernorm.weight"], cfg["rms_norm_eps"], gemma)
        gate = _lin(params, p + "mlp.gate_proj", h, lora_scale)
        g = (F.gelu(gate, approximate="tanh") if gemma else F.silu(gate)) * _lin(params, p + "mlp.up_proj", h, lora_scale)
        x = x + _lin(params, p + "mlp.down_proj", g, lora_scale)
    x = rms_norm(x, params["model.norm.weight"], cfg["rms_norm_eps"], gemma)
    return F.linear(x, emb).float()


def greedy_generate(params, cfg, input_ids, attention_mask, inv_freq, max_new_tokens, eos_token_id, pad_token_id):
    """GenerationMixin greedy search (generation/utils.py:3131, do_sample=False) without a cache: re-runs the
    full forward per step; position ids as HF derives them from the mask (cumsum - 1)."""
    ids, mask = input_ids.clone(), attention_mask.clone()
    done = torch.zeros(ids.shape[0], dtype=torch.bool, device=ids.device)
    for _ in range(max_new_tokens):
        pos = (torch.cumsum(mask, 1) - 1).long()
        pos.masked_fill_(mask == 0, 1)                      # generation/utils.py:410-411
        nxt = llama_logits(params, cfg, ids, mask, pos, inv_freq)[:, -1].argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad_token_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], 1)
        mask = torch.cat([mask, torch.ones_like(mask[:, :1])], 1)
        done = done | (nxt == eos_token_id)
        if bool(done.all()):
            break
    return ids


def random_params(cfg, seed=0, dtype=torch.float32, device="cpu", std=0.02):
    g = torch.Generator().manual_seed(seed)
    H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
    Hq, Hkv, V = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["vocab_size"]
    r = lambda *s: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).to(dtype).to(device)
    gemma = cfg.get("model_type", "llama") == "gemma"      # Gemma's norm weights are offsets from 1
    one = 0.0 if gemma else 1.0
    p = {"model.embed_tokens.weight": r(V, H), "model.norm.weight": torch.full((H,), one, dtype=dtype, device=device)}
    for i in range(cfg["num_hidden_layers"]):
        q = f"model.layers.{i}."
        p[q + "self_attn.q_proj.weight"] = r(Hq * D, H)
        p[q + "self_attn.k_proj.weight"] = r(Hkv * D, H)
        p[q + "self_attn.v_proj.weight"] = r(Hkv * D, H)
        p[q + "self_attn.o_proj.weight"] = r(H, Hq * D)
        p[q + "mlp.gate_proj.weight"] = r(I, H)
        p[q + "mlp.up_proj.weight"] = r(I, H)
        p[q + "mlp.down_proj.weight"] = r(H, I)
        p[q + "input_layernorm.weight"] = (one + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16).to(dtype).to(device)
        p[q + "post_attention_layernorm.weight"] = (one + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16).to(dtype).to(device)
    return p


def llama3_inv_freq(head_dim, theta, scaling):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    if not scaling:
        return inv_freq
    factor, lo, hi, old = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"], scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv_freq
    out = torch.where(wavelen > old / lo, inv_freq / factor, inv_freq)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * out / factor + smooth * out
    medium = ~(wavelen < old / hi) * ~(wavelen > old / lo)
    return torch.where(medium, smoothed, out)
