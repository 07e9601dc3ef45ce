"""Plain-PyTorch restatement of the reference's GPT-2 decoder step (BASELINE config C1's model) -- TEST INFRASTRUCTURE, NOT
PRODUCT CODE (same rules as oracle/oracle.py; also bench.py's cpu_baseline for the C1 leg).

Follows the vendored transformers 4.46.0.dev0 under /root/reference/transformers/src/transformers:
  Conv1D                  pytorch_utils.py:87-113        (y = x @ W + b with W = [in, out])
  GPT2SdpaAttention       models/gpt2/modeling_gpt2.py:458-562, mask semantics of GPT2Attention._attn 183-221
                          (causal AND key not padded; scale 1 / sqrt(head_dim))
  GPT2MLP                 models/gpt2/modeling_gpt2.py:565-579 (gelu_new)
  GPT2Block               models/gpt2/modeling_gpt2.py:585-661 (pre-LayerNorm residual block)
  GPT2Model.forward       wte[input_ids] + wpe[position_ids], ln_f
  GPT2LMHeadModel loss    models/gpt2/modeling_gpt2.py:1300-1304: shift, CrossEntropyLoss on the logits AS THEY ARE (no fp32 upcast)
All dropouts are taken as 0 (eval / parity runs).  Pinned against outputs of the reference itself:
tests/golden/decoder_gpt2_tiny.npz (tests/test_oracle_decoder.py, fp32, 1e-5)."""
import math

import torch
import torch.nn.functional as F


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _conv1d(params, name, x):
    return x @ params[name + ".weight"] + params[name + ".bias"]


def gpt2_logits(params, cfg, input_ids, attention_mask, position_ids):
    """Logits [B, S, V] of GPT2LMHeadModel.forward in the parameters' dtype."""
    H, nh, eps = cfg["n_embd"], cfg["n_head"], cfg.get("layer_norm_epsilon", 1e-5)
    D = H // nh
    B, S = input_ids.shape
    wte = params["transformer.wte.weight"]
    dt = wte.dtype
    x = wte[input_ids] + params["transformer.wpe.weight"][position_ids]
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool, device=x.device))
    visible = causal[None, None] & (attention_mask[:, None, None, :] != 0)
    bias = torch.zeros(B, 1, S, S, dtype=dt, device=x.device).masked_fill(~visible, torch.finfo(dt).min)
    for i in range(cfg["n_layer"]):
        p = f"transformer.h.{i}."
        h = F.layer_norm(x, (H,), params[p + "ln_1.weight"], params[p + "ln_1.bias"], eps)
        q, k, v = _conv1d(params, p + "attn.c_attn", h).split(H, dim=2)
        q, k, v = (t.view(B, S, nh, D).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
        a = a.transpose(1, 2).reshape(B, S, H)
        x = x + _conv1d(params, p + "attn.c_proj", a)
        h = F.layer_norm(x, (H,), params[p + "ln_2.weight"], params[p + "ln_2.bias"], eps)
        x = x + _conv1d(params, p + "mlp.c_proj", gelu_new(_conv1d(params, p + "mlp.c_fc", h)))
    x = F.layer_norm(x, (H,), params["transformer.ln_f.weight"], params["transformer.ln_f.bias"], eps)
    return F.linear(x, wte)


def gpt2_loss(params, cfg, input_ids, attention_mask, labels, position_ids):
    logits = gpt2_logits(params, cfg, input_ids, attention_mask, position_ids)
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=-100)


def random_params(cfg, seed=0, dtype=torch.float32, device="cpu", std=0.02):
    """GPT2PreTrainedModel._init_weights (modeling_gpt2.py:676-702): N(0, initializer_range), c_proj scaled by 1/sqrt(2 n_layer),
    LayerNorm ones/zeros, biases zero -- values rounded to bf16 so that a bf16 model can hold them exactly."""
    g = torch.Generator().manual_seed(seed)
    H, L, V, P = cfg["n_embd"], cfg["n_layer"], cfg["vocab_size"], cfg["n_positions"]
    I = cfg.get("n_inner") or 4 * H
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * std * sc).to(torch.bfloat16).to(dtype).to(device)
    one = lambda n: torch.ones(n, dtype=dtype, device=device)
    zero = lambda n: torch.zeros(n, dtype=dtype, device=device)
    p = {"transformer.wte.weight": r(V, H), "transformer.wpe.weight": r(P, H), "transformer.ln_f.weight": one(H), "transformer.ln_f.bias": zero(H)}
    for i in range(L):
        q = f"transformer.h.{i}."
        p[q + "ln_1.weight"], p[q + "ln_1.bias"], p[q + "ln_2.weight"], p[q + "ln_2.bias"] = one(H), zero(H), one(H), zero(H)
        p[q + "attn.c_attn.weight"], p[q + "attn.c_attn.bias"] = r(H, 3 * H), zero(3 * H)
        p[q + "attn.c_proj.weight"], p[q + "attn.c_proj.bias"] = r(H, H, sc=1.0 / math.sqrt(2 * L)), zero(H)
        p[q + "mlp.c_fc.weight"], p[q + "mlp.c_fc.bias"] = r(H, I), zero(I)
        p[q + "mlp.c_proj.weight"], p[q + "mlp.c_proj.bias"] = r(I, H, sc=1.0 / math.sqrt(2 * L)), zero(H)
    return p
