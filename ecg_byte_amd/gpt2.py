"""GPT-2 on the MI355X kernels of libecgbyte_hip.so: BASELINE config C1's model ("PTB-XL 100 Hz ... GPT-2-small", SURVEY.md section 8d;
the reference loads it like any other causal LM, ecg_byte/main.py:141-158).

`HipGPT2LM` presents the same HuggingFace surface as `decoder.HipCausalLM` (forward with labels / loss.backward(), logits without
labels, generate, resize_token_embeddings, HF-named state_dict, from_pretrained / save_pretrained) and runs the GPT-2 block of the
vendored transformers (models/gpt2/modeling_gpt2.py:183-221,458-661,1300-1304; pytorch_utils.py:87-113):
    x = wte[ids] + wpe[position_ids];  per layer  x += c_proj(attn(c_attn(ln_1(x))));  x += c_proj(gelu_new(c_fc(ln_2(x))));  ln_f;
    tied lm_head; mean CE over labels != -100.
What differs from the Llama block: LayerNorm with bias (ecgb_layernorm_fwd/bwd), learned positions, biased projections (Conv1D stores
[in, out]; here every weight is [out, in] so that all GEMMs are the K-contiguous NT kernel, state_dict transposes), a plain gelu_new MLP,
no RoPE, multi-head attention (Hkv = Hq) on the same fused causal + left-padding attention kernels (head_dim 64 in every GPT-2 size).
The loss: the reference does NOT upcast GPT-2's logits (modeling_gpt2.py:1300-1304 vs loss_utils.py:36): its bf16 run rounds the
log-probabilities to bf16; the kernel here accumulates the same bf16 logits in fp32 (the difference is inside the 1e-2 loss tolerance).
Dropout: embd_pdrop / resid_pdrop / attn_pdrop are applied in training mode (counter-based masks, replayed in the backward).  The
fused attention kernels have no dropout on the probabilities, so a training step with attn_pdrop > 0 (GPT2Config's default 0.1)
takes the materialised-scores attention (head-batched GEMMs + softmax kernel + dropout on P): slower, same semantics
(modeling_gpt2.py:183-221); eval, generation and attn_pdrop = 0 run fused."""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import decoder_ops as ops
from .decoder import HipCausalLM


@dataclass
class GPT2Config:
    """GPT2Config of the vendored transformers (models/gpt2/configuration_gpt2.py:139-149 defaults = GPT-2 small)."""
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    n_inner: int | None = None
    layer_norm_epsilon: float = 1e-5
    resid_pdrop: float = 0.1
    embd_pdrop: float = 0.1
    attn_pdrop: float = 0.1
    initializer_range: float = 0.02
    pad_token_id: int | None = None
    tie_word_embeddings: bool = True
    model_type: str = "gpt2"

    # the names the shared machinery of HipCausalLM reads
    @property
    def hidden_size(self): return self.n_embd
    @property
    def num_hidden_layers(self): return self.n_layer
    @property
    def num_attention_heads(self): return self.n_head
    @property
    def num_key_value_heads(self): return self.n_head
    @property
    def head_dim(self): return self.n_embd // self.n_head
    @property
    def intermediate_size(self): return self.n_inner if self.n_inner is not None else 4 * self.n_embd


class HipGPT2LM(HipCausalLM):
    supports_optimizer_overlap = False   # its forward / backward carry no per-group waits: the optimizer step stays in-stream

    def __init__(self, cfg: GPT2Config, device="cuda", seed: int = 0):
        nn.Module.__init__(self)
        self.cfg = cfg
        self.config = SimpleNamespace(hidden_size=cfg.n_embd, pad_token_id=cfg.pad_token_id, vocab_size=cfg.vocab_size)
        H, I, L = cfg.n_embd, cfg.intermediate_size, cfg.n_layer
        assert H % 64 == 0 and I % 64 == 0 and cfg.head_dim == 64, "GEMM K-step is 64; the fused attention kernels take head_dim 64 (every GPT-2 size)"
        dev = torch.device(device)
        g = torch.Generator(device="cpu").manual_seed(seed)

        def init(*shape, sc=1.0):   # GPT2PreTrainedModel._init_weights, modeling_gpt2.py:676-702
            return (torch.randn(*shape, generator=g) * cfg.initializer_range * sc).to(torch.bfloat16).to(dev)

        def vec(n, v):
            return nn.Parameter(torch.full((n,), v, dtype=torch.bfloat16, device=dev))

        self.v_pad = (cfg.vocab_size + 127) // 128 * 128
        emb = torch.zeros(self.v_pad, H, dtype=torch.bfloat16, device=dev)
        emb[: cfg.vocab_size] = init(cfg.vocab_size, H)
        self.embed = nn.Parameter(emb)
        self.wpe = nn.Parameter(init(cfg.n_positions, H))
        pl = lambda f: nn.ParameterList([f() for _ in range(L)])
        self.ln1, self.ln1_b = pl(lambda: vec(H, 1.0)), pl(lambda: vec(H, 0.0))
        self.wqkv, self.bqkv = pl(lambda: nn.Parameter(init(3 * H, H))), pl(lambda: vec(3 * H, 0.0))
        self.wo, self.bo = pl(lambda: nn.Parameter(init(H, H, sc=1.0 / math.sqrt(2 * L)))), pl(lambda: vec(H, 0.0))
        self.ln2, self.ln2_b = pl(lambda: vec(H, 1.0)), pl(lambda: vec(H, 0.0))
        self.wfc, self.bfc = pl(lambda: nn.Parameter(init(I, H))), pl(lambda: vec(I, 0.0))
        self.wproj, self.bproj = pl(lambda: nn.Parameter(init(H, I, sc=1.0 / math.sqrt(2 * L)))), pl(lambda: vec(H, 0.0))
        self.norm, self.norm_b = vec(H, 1.0), vec(H, 0.0)
        self._anchor = nn.Parameter(torch.zeros(1, device=dev))
        self.gemma = False
        self.embed_scale = 1.0
        self.qkv = 3 * H
        self.lora = None
        self._t, self._t_version = {}, {}
        self.embed_grad_head = None
        self._head_buf = None
        self.full_logits = False
        self._saved = None
        self.grad_sync = None
        self._gflat = self._gflat_key = None
        self.fused_attention = True
        self.drop_seed = seed * 1000003 + 17
        self.drop_calls = 0

    # ---- HF-style surface -------------------------------------------------------------------------------------------
    def enable_lora(self, *a, **k):
        raise NotImplementedError("LoRA targets q_proj .. down_proj (ecg_byte/main.py:136); GPT-2's modules are c_attn / c_proj / c_fc")

    def _hf_named(self):
        """(HF name, tensor, transposed?) -- Conv1D weights are stored [in, out] by the reference (pytorch_utils.py:96)."""
        c = self.cfg
        yield "transformer.wte.weight", self.embed.data[: c.vocab_size], False
        yield "transformer.wpe.weight", self.wpe.data, False
        for i in range(c.n_layer):
            p = f"transformer.h.{i}."
            yield p + "ln_1.weight", self.ln1[i].data, False
            yield p + "ln_1.bias", self.ln1_b[i].data, False
            yield p + "attn.c_attn.weight", self.wqkv[i].data, True
            yield p + "attn.c_attn.bias", self.bqkv[i].data, False
            yield p + "attn.c_proj.weight", self.wo[i].data, True
            yield p + "attn.c_proj.bias", self.bo[i].data, False
            yield p + "ln_2.weight", self.ln2[i].data, False
            yield p + "ln_2.bias", self.ln2_b[i].data, False
            yield p + "mlp.c_fc.weight", self.wfc[i].data, True
            yield p + "mlp.c_fc.bias", self.bfc[i].data, False
            yield p + "mlp.c_proj.weight", self.wproj[i].data, True
            yield p + "mlp.c_proj.bias", self.bproj[i].data, False
        yield "transformer.ln_f.weight", self.norm.data, False
        yield "transformer.ln_f.bias", self.norm_b.data, False
        yield "lm_head.weight", self.embed.data[: c.vocab_size], False

    def state_dict(self, *a, **k):
        return {n: (t.t().contiguous() if tr else t.clone()) for n, t, tr in self._hf_named()}

    def load_state_dict(self, sd, strict=True):
        names = {n: (t, tr) for n, t, tr in self._hf_named()}
        ignore = (".attn.bias", ".attn.masked_bias")                       # the causal-mask buffers old GPT-2 checkpoints carry
        missing = [n for n in names if n not in sd and n != "lm_head.weight"]
        unexpected = [n for n in sd if n not in names and not n.endswith(ignore)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for n, (t, tr) in names.items():
                if n in sd and n != "lm_head.weight":
                    src = sd[n].to(device=t.device, dtype=t.dtype)
                    t.copy_(src.t() if tr else src)
        self._t.clear()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    @classmethod
    def from_pretrained(cls, path, device="cuda", torch_dtype=None, **_):
        import json
        import os
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        keys = ("vocab_size", "n_positions", "n_embd", "n_layer", "n_head", "n_inner", "layer_norm_epsilon", "resid_pdrop", "embd_pdrop",
                "attn_pdrop", "initializer_range", "pad_token_id")
        model = cls(GPT2Config(**{k: hf[k] for k in keys if k in hf}), device=device)
        sd = load_file(os.path.join(path, "model.safetensors"))
        sd = {(n if n.startswith(("transformer.", "lm_head.")) else "transformer." + n): t for n, t in sd.items()}   # the hub's gpt2 has no prefix
        model.load_state_dict(sd)
        return model

    def save_pretrained(self, path):
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        c = self.cfg
        cfg = {"architectures": ["GPT2LMHeadModel"], "model_type": "gpt2", "torch_dtype": "bfloat16", "activation_function": "gelu_new",
               **{k: getattr(c, k) for k in ("vocab_size", "n_positions", "n_embd", "n_layer", "n_head", "n_inner", "layer_norm_epsilon",
                                             "resid_pdrop", "embd_pdrop", "attn_pdrop", "initializer_range", "pad_token_id")}}
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        save_file({n: t.contiguous().cpu() for n, t in self.state_dict().items() if n != "lm_head.weight"},
                  os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    def _grad_groups(self, frozen):
        L = self.cfg.n_layer
        groups = [[self.wqkv[i], self.bqkv[i], self.wo[i], self.bo[i], self.wfc[i], self.bfc[i], self.wproj[i], self.bproj[i],
                   self.ln1[i], self.ln1_b[i], self.ln2[i], self.ln2_b[i]] for i in reversed(range(L))]
        groups.append([self.embed, self.wpe, self.norm, self.norm_b])
        return groups

    # ---- the block ------------------------------------------------------------------------------------------------------
    def _drop(self, t, p, tag):
        """Inverted dropout in training mode; returns (tensor, seed or None).  The same seed replays the mask on the gradient."""
        if not (self.training and p > 0):
            return t, None
        self.drop_calls += 1
        seed = self.drop_seed + 7919 * self.drop_calls + tag
        return ops.dropout(t, p, seed, out=t), seed

    def _embed(self, input_ids, position_ids):
        x = ops.embed_fwd(input_ids.view(-1), self.embed.data, 1.0)
        return ops.add(x, ops.embed_fwd(position_ids.reshape(-1).contiguous(), self.wpe.data, 1.0), out=x)

    def _prep(self, input_ids, attention_mask, position_ids):
        dev = self.device
        ops.require_current(dev)
        input_ids = input_ids.to(dev).contiguous()
        B, S = input_ids.shape
        mask = (attention_mask.to(dev).float() if attention_mask is not None else torch.ones(B, S, device=dev)).contiguous()
        if position_ids is None:
            position_ids = torch.arange(S, device=dev)[None].expand(B, S)
        return input_ids, mask, position_ids.to(dev).long(), B, S

    def _forward_loss(self, input_ids, attention_mask, labels, position_ids):
        c = self.cfg
        H, nh, D, eps = c.n_embd, c.n_head, c.head_dim, c.layer_norm_epsilon
        input_ids, mask, position_ids, B, S = self._prep(input_ids, attention_mask, position_ids)
        labels = labels.to(self.device)
        if S % 64:   # as HipCausalLM._forward_loss: masked, unlabelled positions on the left change nothing
            lpad = 64 - S % 64
            dev = self.device
            input_ids = torch.cat([torch.zeros((B, lpad), dtype=input_ids.dtype, device=dev), input_ids], 1).contiguous()
            mask = torch.cat([torch.zeros((B, lpad), device=dev), mask], 1).contiguous()
            position_ids = torch.cat([torch.zeros((B, lpad), dtype=position_ids.dtype, device=dev), position_ids], 1)
            labels = torch.cat([torch.full((B, lpad), -100, dtype=labels.dtype, device=dev), labels], 1)
            S += lpad
        scale = 1.0 / math.sqrt(D)
        x = self._embed(input_ids, position_ids)
        x, seed_e = self._drop(x, c.embd_pdrop, 1)
        saved, delta = [], None
        for i in range(c.n_layer):
            h1, mu1, rs1, x1 = ops.layernorm_fwd(x, self.ln1[i].data, self.ln1_b[i].data, eps, residual=delta)
            qkv = ops.bias_(ops.gemm_nt(h1, self.wqkv[i].data), self.bqkv[i].data)
            if self.training and c.attn_pdrop > 0:                         # dropout on the probabilities: materialised scores
                self.drop_calls += 1
                adrop = (c.attn_pdrop, self.drop_seed + 7919 * self.drop_calls + 4)
                ao, lse = self._attn_materialised(qkv, mask, B, S, drop=adrop)
                lse = (lse, adrop)
            else:
                ao, lse = ops.attn_fwd(qkv, mask, B, S, nh, nh, D, scale)
            attn_delta = ops.bias_(ops.gemm_nt(ao, self.wo[i].data), self.bo[i].data)
            attn_delta, seed_a = self._drop(attn_delta, c.resid_pdrop, 2)
            h2, mu2, rs2, x2 = ops.layernorm_fwd(x1, self.ln2[i].data, self.ln2_b[i].data, eps, residual=attn_delta)
            u = ops.gemm_nt(h2, self.wfc[i].data)
            hm = ops.bias_gelu_new_(u, self.bfc[i].data)                   # u now holds the pre-activation
            delta = ops.bias_(ops.gemm_nt(hm, self.wproj[i].data), self.bproj[i].data)
            delta, seed_m = self._drop(delta, c.resid_pdrop, 3)
            saved.append((x1, mu1, rs1, h1, qkv, lse, ao, x2, mu2, rs2, h2, u, hm, seed_a, seed_m))
            x = x2
        hf, muf, rsf, xf = ops.layernorm_fwd(x, self.norm.data, self.norm_b.data, eps, residual=delta)
        loss, dhf = self._loss_head(hf, labels, B, S)
        self._saved = (saved, input_ids, position_ids, mask, (xf, muf, rsf), dhf, (B, S), seed_e)
        return loss.squeeze(0)

    def _backward(self, grad_out):
        if self.grad_sync is not None and hasattr(self.grad_sync, "backward_kernels"):
            with self.grad_sync.backward_kernels():     # one-tile input-gradient GEMMs only while a gradient exchange can be in flight
                return self._backward_impl(grad_out)
        return self._backward_impl(grad_out)

    def _backward_impl(self, grad_out):
        c = self.cfg
        H, nh, D = c.n_embd, c.n_head, c.head_dim
        saved, input_ids, position_ids, mask, (xf, muf, rsf), dhf, (B, S), seed_e = self._saved
        self._saved = None
        dev = self.device
        scale = 1.0 / math.sqrt(D)
        go = float(grad_out)
        if go != 1.0:
            dhf = (dhf.float() * go).to(torch.bfloat16)
            self.embed_grad_head.copy_((self.embed_grad_head.float() * go).to(torch.bfloat16))
        self._grad_layout()
        L = c.n_layer
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)

        def undrop(t, p, seed):
            return t if seed is None else ops.dropout(t, p, seed)

        dw, db = z(H), z(H)
        g = ops.layernorm_bwd(xf, self.norm.data, muf, rsf, dhf, dw, db)
        self._vgrad(self.norm, dw)
        self._vgrad(self.norm_b, db)
        for i in reversed(range(L)):
            x1, mu1, rs1, h1, qkv, lse, ao, x2, mu2, rs2, h2, u, hm, seed_a, seed_m = saved.pop()
            d_delta = undrop(g, c.resid_pdrop, seed_m)
            self._vgrad(self.bproj[i], ops.colsum(d_delta))
            self._wgrad(self.wproj[i], d_delta, hm)
            d_hm = self._dx(d_delta, ("wproj", i), self.wproj[i])
            d_u = ops.gelu_new_bwd(u, d_hm)
            del d_hm, hm, u
            self._vgrad(self.bfc[i], ops.colsum(d_u))
            self._wgrad(self.wfc[i], d_u, h2)
            d_h2 = self._dx(d_u, ("wfc", i), self.wfc[i])
            del d_u
            dw, db = z(H), z(H)
            g2 = ops.layernorm_bwd(x2, self.ln2[i].data, mu2, rs2, d_h2, dw, db, dres=g)
            self._vgrad(self.ln2[i], dw)
            self._vgrad(self.ln2_b[i], db)
            d_ad = undrop(g2, c.resid_pdrop, seed_a)
            self._vgrad(self.bo[i], ops.colsum(d_ad))
            self._wgrad(self.wo[i], d_ad, ao)
            d_ao = self._dx(d_ad, ("wo", i), self.wo[i])
            if isinstance(lse, tuple):
                d_qkv = self._attn_materialised_bwd(qkv, d_ao, lse[0], B, S, drop=lse[1])
            else:
                d_qkv = ops.attn_bwd(qkv, mask, ao, d_ao, lse, B, S, nh, nh, D, scale)
            self._vgrad(self.bqkv[i], ops.colsum(d_qkv))
            self._wgrad(self.wqkv[i], d_qkv, h1)
            d_h1 = self._dx(d_qkv, ("wqkv", i), self.wqkv[i])
            dw, db = z(H), z(H)
            g = ops.layernorm_bwd(x1, self.ln1[i].data, mu1, rs1, d_h1, dw, db, dres=g2)
            self._vgrad(self.ln1[i], dw)
            self._vgrad(self.ln1_b[i], db)
            if self.grad_sync is not None:
                self.grad_sync.on_flat_ready(self._gflat, *self._granges[L - 1 - i])
        g = undrop(g, c.embd_pdrop, seed_e)
        # masked rows that carry no loss: gradient exactly zero -- skipped instead of summed by one workgroup as one long run (a masked row with a valid shifted
        # label keeps its gradient, as autograd gives it: GPT-2 has no padding_idx)
        dead = (mask.reshape(-1) == 0) & self._unlabelled_rows
        self._embedding_grad(self.embed, self.embed_grad_head, input_ids.view(-1).masked_fill(dead, -1), g, 1.0, -1)     # GPT-2's wte has no padding_idx
        self._embedding_grad(self.wpe, None, position_ids.reshape(-1).masked_fill(dead, -1).contiguous(), g, 1.0, -1)
        if self.grad_sync is not None:
            self.grad_sync.on_flat_ready(self._gflat, *self._granges[L])
            self.grad_sync.finish()

    # ---- inference ------------------------------------------------------------------------------------------------------
    def _layer_infer(self, i, x, delta, attend):
        c = self.cfg
        eps = c.layer_norm_epsilon
        h1, _, _, x = ops.layernorm_fwd(x, self.ln1[i].data, self.ln1_b[i].data, eps, residual=delta)
        qkv = ops.bias_(ops.gemm_nt(h1, self.wqkv[i].data), self.bqkv[i].data)
        ao = attend(i, qkv)
        attn_delta = ops.bias_(ops.gemm_nt(ao, self.wo[i].data), self.bo[i].data)
        h2, _, _, x = ops.layernorm_fwd(x, self.ln2[i].data, self.ln2_b[i].data, eps, residual=attn_delta)
        hm = ops.bias_gelu_new_(ops.gemm_nt(h2, self.wfc[i].data), self.bfc[i].data)
        return x, ops.bias_(ops.gemm_nt(hm, self.wproj[i].data), self.bproj[i].data)

    def _hidden_states(self, input_ids, attention_mask=None, position_ids=None, kv_out=None):
        c = self.cfg
        H, nh, D = c.n_embd, c.n_head, c.head_dim
        input_ids, mask, position_ids, B, S = self._prep(input_ids, attention_mask, position_ids)
        scale = 1.0 / math.sqrt(D)

        def attend(i, qkv):
            if kv_out is not None:
                kv_out[i][:, :S].copy_(qkv.view(B, S, 3 * H)[:, :, H:])
            return ops.attn_fwd(qkv, mask, B, S, nh, nh, D, scale)[0]

        x, delta = self._embed(input_ids, position_ids), None
        for i in range(c.n_layer):
            x, delta = self._layer_infer(i, x, delta, attend)
        return ops.layernorm_fwd(x, self.norm.data, self.norm_b.data, c.layer_norm_epsilon, residual=delta)[0]

    def _decode_step(self, tokens, pos, mask, caches, n, n_dev=None, scratch=None):
        c = self.cfg
        H, nh, D = c.n_embd, c.n_head, c.head_dim
        scale = 1.0 / math.sqrt(D)

        def attend(i, qkv):
            ns = ops.decode_splits(caches[i].shape[1], qkv.shape[0], nh)      # by the caches' capacity: see HipCausalLM._decode_step
            if n_dev is None:
                caches[i][:, n - 1].copy_(qkv[:, H:])
                if ns > 1:
                    return ops.attn_decode_split(qkv, caches[i], mask, n, nh, nh, D, scale, ns)
                return ops.attn_decode(qkv, caches[i], mask, n, nh, nh, D, scale)
            ops.kv_append(qkv, H, caches[i], n_dev)
            if ns > 1:
                return ops.attn_decode_split(qkv, caches[i], mask, n_dev, nh, nh, D, scale, ns, scratch=scratch)
            return ops.attn_decode_dyn(qkv, caches[i], mask, n_dev, nh, nh, D, scale)

        x, delta = self._embed(tokens[:, None], pos[:, None]), None
        for i in range(c.n_layer):
            x, delta = self._layer_infer(i, x, delta, attend)
        return ops.layernorm_fwd(x, self.norm.data, self.norm_b.data, c.layer_norm_epsilon, residual=delta)[0]

    def _rope_tables(self, position_ids):   # (generate() of the base class never needs them here)
        raise NotImplementedError

    def resize_token_embeddings(self, n: int):
        """As HipCausalLM.resize_token_embeddings (modeling_utils.py:2080-2176, new rows = the mean row)."""
        old = self.cfg.vocab_size
        if n == old:
            return
        H = self.cfg.n_embd
        v_pad = (n + 127) // 128 * 128
        emb = torch.zeros(v_pad, H, dtype=torch.bfloat16, device=self.device)
        keep = min(old, n)
        emb[:keep] = self.embed.data[:keep]
        if n > old:
            emb[old:n] = self.embed.data[:old].float().mean(0).to(torch.bfloat16)
        self.embed = nn.Parameter(emb)
        self.cfg.vocab_size = self.config.vocab_size = n
        self.v_pad = v_pad
        self.embed_grad_head = None
        self._head_buf = None
        self._t.pop("embed", None)
        self._gflat = None
