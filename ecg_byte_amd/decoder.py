"""Causal-LM training step on the MI355X kernels of libecgbyte_hip.so -- the model side of the
reference's hot path.

`HipCausalLM` presents the slice of the HuggingFace surface that ecg_byte/main.py:141-165 and
ecg_byte/models/llm.py:17-24 touch:
    out = llm(input_ids=..., attention_mask=..., labels=..., position_ids=...);  out.loss.backward()
    llm.config.{hidden_size, pad_token_id}, llm.device, llm.resize_token_embeddings(n),
    state_dict()/load_state_dict() with HF parameter names (checkpoints, main.py:193-195,299-306)
and runs Llama-architecture decoders (vendored transformers/models/llama/modeling_llama.py:859-1225):
pre-norm blocks with RMSNorm, half-split RoPE (default or llama3 frequencies), grouped-query
causal attention with the reference's left-padding mask, SwiGLU MLP, tied lm_head, mean CE over
labels != -100.  Every tensor op is a kernel of include/ecgbyte_decoder.h; torch supplies
parameters, buffers, streams and (for N > 1) RCCL all-reduce.

Layout choices (MI355X): q/k/v and gate/up projections are fused into one weight each (one big
GEMM instead of three / two); every weight keeps a transposed shadow copy so that all three GEMMs
of a linear layer (y = xW^T, dx = dy W, dW = dy^T x) are the same K-contiguous NT kernel; the
vocabulary is padded to a multiple of 128 rows; the loss head only materialises logits for rows
whose (shifted) label is not -100 -- identical loss and gradients, a fraction of the work.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import decoder_ops as ops


@dataclass
class DecoderConfig:
    vocab_size: int = 128256
    hidden_size: int = 2048
    intermediate_size: int = 8192
    num_hidden_layers: int = 16
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int | None = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: dict | None = field(default_factory=lambda: {
        "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
        "original_max_position_embeddings": 8192, "rope_type": "llama3"})
    tie_word_embeddings: bool = True
    pad_token_id: int | None = None
    initializer_range: float = 0.02
    model_type: str = "llama"      # "gemma": RMSNorm with (1 + w) in fp32 (modeling_gemma.py:60-65), gelu-tanh gate (140-149),
                                   # embeddings scaled by sqrt(hidden) in the activation dtype (800-801), head_dim from the config

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads

    @staticmethod
    def gemma_2b(**kw):
        """Published Gemma-2B dimensions (config C5): 18 layers, hidden 2048, 8 query heads / 1 KV head of 256, MLP 16384."""
        base = dict(vocab_size=256000, hidden_size=2048, intermediate_size=16384, num_hidden_layers=18, num_attention_heads=8,
                    num_key_value_heads=1, head_dim=256, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, model_type="gemma")
        base.update(kw)
        return DecoderConfig(**base)

    @staticmethod
    def llama_3_2_1b(**kw):
        """Published Llama-3.2-1B dimensions."""
        return DecoderConfig(**kw)


def rope_inv_freq(cfg: DecoderConfig) -> torch.Tensor:
    """_compute_default_rope_parameters / _compute_llama3_parameters
    (transformers/src/transformers/modeling_rope_utils.py:29-68,310-351)."""
    dim = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    rs = cfg.rope_scaling
    if not rs or rs.get("rope_type", rs.get("type", "default")) == "default":
        return inv_freq
    if rs.get("rope_type", rs.get("type")) != "llama3":
        raise NotImplementedError(f"rope_type {rs.get('rope_type')}")
    factor, lo, hi, old = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"], rs["original_max_position_embeddings"]
    low_freq_wavelen, high_freq_wavelen = old / lo, old / hi
    wavelen = 2 * math.pi / inv_freq
    inv_freq_llama = torch.where(wavelen > low_freq_wavelen, inv_freq / factor, inv_freq)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * inv_freq_llama / factor + smooth * inv_freq_llama
    is_medium = ~(wavelen < high_freq_wavelen) * ~(wavelen > low_freq_wavelen)
    return torch.where(is_medium, smoothed, inv_freq_llama)


def _dp_rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class LoraSite(nn.Module):
    """LoRA adapters of one (possibly fused) projection, as peft's LoraLayer computes them
    (ecg_byte/main.py:131-155: r 16, alpha 32, dropout 0.05 on q,k,v,o,gate,up,down):
        y[:, block b] = x W_b^T + (alpha / r) * (dropout_b(x) A_b^T) B_b^T,   A_b: [r, in] kaiming-uniform, B_b: [out_b, r] zeros,
    with an independent dropout mask per module (q, k and v are three LoraLayers over the same input).
    peft is neither vendored nor installed: parity unpinned, restated from the published formula.

    MI355X layout: the adapters of the site are STACKED -- A is [64, in] (block b's rows start at 16 * spb * b, spb = ceil(r / 16)
    sixteen-row sub-blocks per block; the rows past r and past the last block are zero and stay zero), B is [out, 64]
    block-structured (block b's rows carry its r columns, zeros elsewhere: the gradient is masked to that pattern).  Then
      forward   t = lora_down(x)  one pass over x (dropout inside, ecgb_lora_down);  y = [x | t] . [W | B]^T  ONE launch (K-concatenated
                GEMM): the branch costs one extra K-step of the projection instead of a read-modify-write of y
      backward  dt = dy . B;  dB = dy^T . t;  dA_b = dt_b^T . (mask_b . x);  dx += sum_b mask_b . (dt_b A_b)  (ecgb_lora_dx, in place)."""

    def __init__(self, in_dim, out_blocks, r, alpha, dropout, device, gen):
        super().__init__()
        self.r, self.alpha, self.p = r, alpha, dropout
        self.blocks = list(out_blocks)                       # (column offset, width) inside the fused output
        nb = len(self.blocks)
        self.spb = (r + 15) // 16                            # sub-blocks of 16 ranks per block
        self.n_sub, self.n_fields = nb * self.spb, nb
        if self.n_sub > 4:
            raise NotImplementedError(f"LoRA rank {r} on a projection of {nb} fused modules needs {self.n_sub} > 4 sub-blocks of 16")
        self.scale = alpha / r
        A = torch.zeros(64, in_dim)
        bound = 1.0 / math.sqrt(in_dim)                      # kaiming_uniform_(a=sqrt(5)) on [r, in]
        n_out = sum(w for _, w in self.blocks)
        bmask = torch.zeros(n_out, 64)
        for b, (off, w) in enumerate(self.blocks):
            lo = 16 * self.spb * b
            A[lo: lo + r] = (torch.rand(r, in_dim, generator=gen) * 2 - 1) * bound
            bmask[off: off + w, lo: lo + r] = 1.0
        self.A = nn.Parameter(A.to(torch.bfloat16).to(device))
        self.B = nn.Parameter(torch.zeros(n_out, 64, dtype=torch.bfloat16, device=device))
        self.register_buffer("bmask", bmask.to(torch.bfloat16).to(device), persistent=False)
        self.seed = int(torch.randint(0, 2 ** 31, (1,), generator=gen))
        self.calls = 0

    def a_rows(self, b):
        """Rows of A (= columns of B) that hold block b's rank-r adapter."""
        lo = 16 * self.spb * b
        return lo, lo + self.r

    def project(self, x, w, training, keep=False, glu=None, rope=None, t_pre=None, sync=None):
        """y = x W^T + the adapter branch, [T, out].  Returns (y, what backward needs).
        glu = (gelu_tanh, keep_gu): the site is the fused gate|up projection and the GLU runs in the GEMM's epilogue; y is then the pair
        (gate|up or None, act(gate) * up).  sync: the model's decode-step counter (int32[1], a new value every step) -- a decode step's o / down site then forms t and the
        projection in ONE launch (ops.gemm_nt_lora_decode: the same bits)."""
        p, seed = 0.0, 0
        if training and self.p > 0:
            # One mask stream per (site, rank, call): data-parallel ranks draw DIFFERENT masks (torch's per-process generators give the
            # reference's DDP ranks different dropout too); `calls` restarts at 0 with a new process -- the reference never resumes
            # training -- and HipCausalLM.lora_rng_state() / set_lora_rng_state() carry it across a checkpoint.
            self.calls += 1
            p, seed = self.p, (self.seed + 7919 * self.calls + 104729 * _dp_rank()) & 0x7FFFFFFF
        if sync is not None and t_pre is None and p == 0.0 and glu is None and rope is None and ops.lora_decode_ok(x.shape[0], w.shape[0], x.shape[1], self.A.shape[0]):
            if getattr(self, "dec_t64", None) is None or self.dec_t64.device != x.device:
                self.dec_t64 = torch.zeros((2, self.A.shape[0]), dtype=torch.int64, device=x.device)      # (the counter starts at 1: zero is no step's value)
            return ops.gemm_nt_lora_decode(x, w, self.A.data, self.scale, self.B.data, self.dec_t64, sync), None
        if t_pre is not None and p == 0.0:          # a decode step whose norm kernel already formed t = scale * x A^T (ops.rmsnorm_fwd(lora=...))
            t, xd = t_pre, None
        elif x.shape[0] <= 8 and p == 0.0:          # a decode step: the few-row GEMM reads A once at HBM speed
            t, xd = ops.gemm_nt(x, self.A.data, alpha=self.scale), None
        else:
            t, xd = ops.lora_down(x, self.A.data, self.n_sub, self.n_fields, self.scale, p, seed, keep_masked=keep)
        if glu is not None and ops.glu_fusable(x.shape[0], w.shape[0] // 2):
            y = ops.gemm_nt_glu(x, w, gelu_tanh=glu[0], keep_gu=glu[1], a2=t, b2=self.B.data)
        elif rope is not None:                              # the q|k|v site: RoPE in the projection's epilogue (or behind it: ops.gemm_nt_rope decides)
            y = ops.gemm_nt_rope(x, w, rope[0], rope[1], rope[2], a2=t, b2=self.B.data)
        else:
            y = ops.gemm_nt(x, w, a2=t, b2=self.B.data)
            if glu is not None:
                y = (y if glu[1] else None, ops.glu_fwd(y, gelu_tanh=glu[0]))
        return y, (x, xd, t, p, seed)

    def backward(self, dy, saved, model, dx, glu=None, weight=None):
        """Sets A.grad / B.grad (slices of the model's flat gradient buffer) and adds the adapters' contribution to `dx` in place
        (dx already holds the base projection's dy . W).  glu = (gate|up, gelu_tanh) on the down-projection site: dx is d(act(gate) * up);
        returns d(gate|up) from the same pass instead of writing dx back (ecgb_lora_dx_glu).  dx may also be a callable that forms the base
        projection's dy . W when asked, with `weight` the frozen weight itself: on a single-module site (o, down) where the four-wave GEMM takes
        the shape, product and adapter share (and the GLU backward) are ONE launch (ops.gemm_nn_lora / ops.gemm_nn_glu_bwd_lora) and the
        callable is never called."""
        x, xd, t, p, seed = saved
        Bt = model._shadow(("lora_Bt", id(self)), self.B)                 # [64, out]
        At = model._shadow(("lora_At", id(self)), self.A)                 # [in, 64]
        dt = ops.gemm_nt(dy, Bt)                                         # [T, 64] = dy . B
        model._wgrad(self.B, dy, t)                                      # dB = dy^T . t  (t carries alpha / r and 1 / (1 - p))
        if self.n_fields > 1:                                            # every block's rows keep only its own columns (one module: t is zero past its rank, dB with it)
            self.B.grad.mul_(self.bmask)
        keep_scale = self.scale / (1.0 - int(p * 65536.0) / 65536.0)     # the keep probability ecgb_lora_down's 16-bit threshold gives
        if p == 0.0:
            model._wgrad(self.A, dt, x, alpha=keep_scale)                # no dropout: one product for all blocks
        else:                                                            # dA_b = dt_b^T . (mask_b . x), the masks replayed from the seed
            model._lora_agrad(self.A, x, dt, self.n_sub, self.n_fields, self.scale, p, seed)
        if dx is None:                                                   # the caller needs no input gradient (the bottom layer over frozen embeddings)
            return None
        if callable(dx):
            if self.n_sub == 1 and weight is not None:
                if glu is not None:
                    fused = ops.gemm_nn_glu_bwd_lora(dy, weight, glu[0], dt, At, self.scale, p, seed, gelu_tanh=glu[1])
                else:
                    fused = ops.gemm_nn_lora(dy, weight, dt, At, self.scale, p, seed)
                if fused is not None:
                    return fused
            dx = dx()
        if glu is not None and self.n_sub == 1:
            return ops.lora_dx_glu(dx, dt, At, glu[0], self.scale, p, seed, gelu_tanh=glu[1])
        ops.lora_dx_(dx, dt, At, self.n_sub, self.n_fields, self.scale, p, seed)
        return dx if glu is None else ops.glu_bwd(glu[0], dx, gelu_tanh=glu[1])


class _LossFn(torch.autograd.Function):
    """Ties the hand-written forward/backward into autograd so `out.loss.backward()` works."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, attention_mask, labels, position_ids):
        ctx.model = model
        loss = model._forward_loss(input_ids, attention_mask, labels, position_ids)
        # The model keeps ONE forward's state (`_saved`, and the tied head's dE, which the loss head writes during the forward -- into the embedding's slice of the flat
        # gradient buffer when that is free): a second grad-enabled forward before this one's backward replaces both.  Each forward is numbered so that the backward of
        # an overwritten one fails loudly instead of consuming the other forward's head gradient (validation passes belong under torch.no_grad()).
        model._fwd_serial = getattr(model, "_fwd_serial", 0) + 1
        ctx.fwd_serial = model._fwd_serial
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if getattr(ctx.model, "_fwd_serial", None) != ctx.fwd_serial or ctx.model._saved is None:
            raise RuntimeError("backward() of a forward whose saved state a later grad-enabled forward has replaced (the model keeps one forward's activations and "
                               "head gradient): run validation / extra losses under torch.no_grad(), or call backward() before the next forward")
        ctx.model._backward(grad_out)
        return None, None, None, None, None, None


class HipCausalLM(nn.Module):
    # HipAdam(overlap=True) needs the forward to wait per parameter group (`_wait_group`) and backward / inference to `sync_optimizer`: a subclass whose
    # own forward / backward do not (HipGPT2LM) sets this False and gets the plain in-stream step
    supports_optimizer_overlap = True
    decode_attn_one = True      # a decode step's RoPE + append + attention as one launch where the shape allows (ops.decode_one_ok); False: round 4's four launches, the same bits
    decode_lora_one = False     # True: a decode step's o / down adapter sites form t = scale * x A^T inside the projection's launch (ops.gemm_nt_lora_decode: two launches a layer less, the same
                                # bits) -- measured 503 against 508 tokens/s at the C5 shape: two dependent dot products and a hand-over between XCDs take what the launch boundary took
    def __init__(self, cfg: DecoderConfig, device="cuda", seed: int = 0):
        super().__init__()
        self.cfg = cfg
        self.config = SimpleNamespace(hidden_size=cfg.hidden_size, pad_token_id=cfg.pad_token_id, vocab_size=cfg.vocab_size)
        H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        Hq, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
        assert cfg.tie_word_embeddings, "untied lm_head not needed by the reference's models"
        assert H % 64 == 0 and I % 64 == 0 and D % 64 == 0, "GEMM K-step is 64"
        self.qkv = Hq * D + 2 * Hkv * D
        dev = torch.device(device)
        g = torch.Generator(device="cpu").manual_seed(seed)

        def init(*shape):   # LlamaPreTrainedModel._init_weights: N(0, initializer_range), modeling_llama.py:737-746
            return (torch.randn(*shape, generator=g) * cfg.initializer_range).to(torch.bfloat16).to(dev)

        self.v_pad = (cfg.vocab_size + 127) // 128 * 128
        emb = torch.zeros(self.v_pad, H, dtype=torch.bfloat16, device=dev)
        emb[: cfg.vocab_size] = init(cfg.vocab_size, H)
        if cfg.pad_token_id is not None and cfg.pad_token_id < cfg.vocab_size:
            emb[cfg.pad_token_id].zero_()
        self.embed = nn.Parameter(emb)
        self.wqkv = nn.ParameterList([nn.Parameter(init(self.qkv, H)) for _ in range(cfg.num_hidden_layers)])
        self.wo = nn.ParameterList([nn.Parameter(init(H, Hq * D)) for _ in range(cfg.num_hidden_layers)])
        self.wgu = nn.ParameterList([nn.Parameter(init(2 * I, H)) for _ in range(cfg.num_hidden_layers)])
        self.wdown = nn.ParameterList([nn.Parameter(init(H, I)) for _ in range(cfg.num_hidden_layers)])
        self.gemma = cfg.model_type == "gemma"
        if cfg.model_type not in ("llama", "gemma"):
            raise NotImplementedError(f"model_type {cfg.model_type!r}: the Llama and Gemma blocks are built")
        # Gemma multiplies the embeddings by sqrt(hidden) held in the activation dtype (modeling_gemma.py:800-801)
        self.embed_scale = float(torch.tensor(H ** 0.5, dtype=torch.bfloat16)) if self.gemma else 1.0
        ones = lambda: nn.Parameter(torch.full((H,), 0.0 if self.gemma else 1.0, dtype=torch.bfloat16, device=dev))   # Gemma's norm weight is an offset from 1
        self.ln1 = nn.ParameterList([ones() for _ in range(cfg.num_hidden_layers)])
        self.ln2 = nn.ParameterList([ones() for _ in range(cfg.num_hidden_layers)])
        self.norm = ones()
        self.register_buffer("inv_freq", rope_inv_freq(cfg).to(dev), persistent=False)
        self._anchor = nn.Parameter(torch.zeros(1, device=dev))   # keeps the autograd node alive when the base is frozen
        self.lora = None        # nn.ModuleDict of LoraSite per layer once enable_lora() ran
        self._t = {}            # transposed shadow weights
        self._t_version = {}    # parameter version each shadow was made from
        self.embed_grad_head = None
        self._head_buf = None
        self.full_logits = False   # True: run the loss head over every row, as the reference materialises it
        self._saved = None
        self.grad_sync = None      # parallel.GradAllReduce: told as soon as a layer's gradients are final
        self._opt_ready = {}       # HipAdam(overlap=True): parameter group -> event recorded behind its update on the optimizer's side stream
        self._opt_done = None      # ... and behind the last one
        self._gflat = None         # flat bf16 buffer holding every trainable gradient (see _grad_layout)
        self._gflat_key = None
        self.fused_attention = cfg.head_dim in ops.FUSED_HEAD_DIMS   # False: materialised scores (batched GEMM + softmax kernels)

    # ---- HF-style surface -------------------------------------------------------------------
    @property
    def device(self):
        return self.embed.device

    @classmethod
    def from_pretrained(cls, path, device="cuda", torch_dtype=None, **_):
        """`AutoModelForCausalLM.from_pretrained(dir, torch_dtype=torch.bfloat16)` (ecg_byte/main.py:142) for a LOCAL
        Llama checkpoint directory in the hub layout: config.json (LlamaConfig fields) + model.safetensors, or the
        sharded model-0000x-of-0000y.safetensors with model.safetensors.index.json.  Weights are cast to bf16."""
        import json
        import os
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        arch = hf.get("model_type", "llama")
        if arch == "gpt2":
            from .gpt2 import HipGPT2LM
            return HipGPT2LM.from_pretrained(path, device=device)
        if arch not in ("llama", "gemma"):
            raise NotImplementedError(f"from_pretrained: model_type {arch!r}: the Llama, Gemma and GPT-2 blocks are built (DESIGN.md §8)")
        keys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
                "head_dim", "rms_norm_eps", "rope_theta", "rope_scaling", "tie_word_embeddings", "pad_token_id", "initializer_range")
        kw = {k: hf[k] for k in keys if k in hf}
        kw["model_type"] = arch
        kw.setdefault("rope_scaling", None)
        kw.setdefault("tie_word_embeddings", arch == "gemma")   # GemmaConfig ties by default, LlamaConfig does not
        if arch == "gemma":
            kw.setdefault("rms_norm_eps", 1e-6)
            kw.setdefault("rope_theta", 10000.0)
        if "num_key_value_heads" not in kw:
            kw["num_key_value_heads"] = kw["num_attention_heads"]
        model = cls(DecoderConfig(**kw), device=device)
        index = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
        else:
            shards = ["model.safetensors"]
        sd = {}
        for sh in shards:
            sd.update(load_file(os.path.join(path, sh)))
        model.load_state_dict(sd)
        gen = os.path.join(path, "generation_config.json")
        if os.path.exists(gen):                                    # what HF's generate() falls back to when the caller passes no sampling arguments
            with open(gen) as f:
                model.generation_config = json.load(f)
        return model

    def save_pretrained(self, path):
        """config.json + model.safetensors in the hub layout (tied lm_head omitted, as transformers does)."""
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        c = self.cfg
        cfg = {"architectures": ["GemmaForCausalLM" if self.gemma else "LlamaForCausalLM"], "model_type": c.model_type, "torch_dtype": "bfloat16",
               **({"hidden_activation": "gelu_pytorch_tanh", "hidden_act": "gelu_pytorch_tanh"} if self.gemma else {"hidden_act": "silu"}),
               "attention_bias": False, "mlp_bias": False,
               **{k: getattr(c, k) for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                             "num_key_value_heads", "head_dim", "rms_norm_eps", "rope_theta", "rope_scaling",
                                             "tie_word_embeddings", "pad_token_id", "initializer_range")}}
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        save_file({n: t.contiguous().cpu() for n, t in self._hf_named() if n != "lm_head.weight"},
                  os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    def forward(self, input_ids=None, attention_mask=None, labels=None, position_ids=None, output_attentions=False, **_):
        if labels is None:   # inference forward: logits only (no autograd graph)
            with torch.no_grad():
                B, S = input_ids.shape
                hf = self._hidden_states(input_ids, attention_mask, position_ids)
                logits = ops.gemm_nt(hf, self.embed.data)[:, :self.cfg.vocab_size].float().view(B, S, -1)
            return SimpleNamespace(loss=None, logits=logits, attentions=None)
        if not torch.is_grad_enabled():   # validation: loss only, nothing kept for a backward
            return SimpleNamespace(loss=self._eval_loss(input_ids, attention_mask, labels, position_ids), logits=None, attentions=None)
        anchor = self._anchor   # a parameter that requires grad, so autograd records the node
        loss = _LossFn.apply(anchor, self, input_ids, attention_mask, labels, position_ids)
        return SimpleNamespace(loss=loss, logits=None, attentions=None)

    def resize_token_embeddings(self, n: int):
        """transformers/modeling_utils.py:2080-2176, with new rows drawn as modeling_utils.py:2464-2489 does:
        N(mean of the old table, 1e-9 * covariance) -- i.e. the mean row up to ~1e-5 noise."""
        old = self.cfg.vocab_size
        if n == old:
            return
        H = self.cfg.hidden_size
        v_pad = (n + 127) // 128 * 128
        emb = torch.zeros(v_pad, H, dtype=torch.bfloat16, device=self.device)
        keep = min(old, n)
        emb[:keep] = self.embed.data[:keep]
        if n > old:
            mean = self.embed.data[:old].float().mean(0)
            emb[old:n] = mean.to(torch.bfloat16)
        self.embed = nn.Parameter(emb)
        self.cfg.vocab_size = n
        self.config.vocab_size = n
        self.v_pad = v_pad
        self.embed_grad_head = None
        self._head_buf = None
        self._t.pop("embed", None)

    def enable_lora(self, r=16, alpha=32, dropout=0.05, seed=0):
        """get_peft_model(llm, LoraConfig(r, lora_alpha, target_modules=[q,k,v,o,gate,up,down]_proj, lora_dropout))
        (ecg_byte/main.py:131-155): base weights, norms and embeddings are frozen, only the adapters train."""
        c = self.cfg
        H, I, D, Hq, Hkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for p in self.parameters():
            p.requires_grad_(False)
        self._anchor.requires_grad_(True)
        sites = nn.ModuleList()
        for _ in range(c.num_hidden_layers):
            sites.append(nn.ModuleDict({
                "qkv": LoraSite(H, [(0, Hq * D), (Hq * D, Hkv * D), (Hq * D + Hkv * D, Hkv * D)], r, alpha, dropout, self.device, gen),
                "o": LoraSite(Hq * D, [(0, H)], r, alpha, dropout, self.device, gen),
                "gu": LoraSite(H, [(0, I), (I, I)], r, alpha, dropout, self.device, gen),
                "down": LoraSite(I, [(0, H)], r, alpha, dropout, self.device, gen),
            }))
        self.lora = sites
        return self

    def lora_rng_state(self):
        """Dropout stream positions of every adapter site (save beside a checkpoint to continue an interrupted run with the mask sequence
        it would have drawn; the reference has no training resume, so its checkpoint format has no slot for this)."""
        return [[layer[k].calls for k in ("qkv", "o", "gu", "down")] for layer in self.lora] if self.lora is not None else None

    def set_lora_rng_state(self, state):
        for layer, row in zip(self.lora, state):
            for k, n in zip(("qkv", "o", "gu", "down"), row):
                layer[k].calls = int(n)

    def lora_named(self):
        """(peft-style name, tensor) pairs of the adapter weights: lora_A [r, in], lora_B [out, r]."""
        names = {"qkv": ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], "o": ["self_attn.o_proj"],
                 "gu": ["mlp.gate_proj", "mlp.up_proj"], "down": ["mlp.down_proj"]}
        for i, layer in enumerate(self.lora):
            for key, mods in names.items():
                site = layer[key]
                for b, mod in enumerate(mods):
                    pre = f"base_model.model.model.layers.{i}.{mod}."
                    lo, hi = site.a_rows(b)
                    off, w = site.blocks[b]
                    yield pre + "lora_A.default.weight", site.A.data[lo:hi]
                    yield pre + "lora_B.default.weight", site.B.data[off: off + w, lo:hi]

    def _hf_named(self):
        """(HF name, tensor view) pairs, modeling_llama.py parameter names."""
        c = self.cfg
        H, I, D, Hq, Hkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        yield "model.embed_tokens.weight", self.embed.data[: c.vocab_size]
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            w = self.wqkv[i].data
            yield p + "self_attn.q_proj.weight", w[: Hq * D]
            yield p + "self_attn.k_proj.weight", w[Hq * D: Hq * D + Hkv * D]
            yield p + "self_attn.v_proj.weight", w[Hq * D + Hkv * D:]
            yield p + "self_attn.o_proj.weight", self.wo[i].data
            yield p + "mlp.gate_proj.weight", self.wgu[i].data[:I]
            yield p + "mlp.up_proj.weight", self.wgu[i].data[I:]
            yield p + "mlp.down_proj.weight", self.wdown[i].data
            yield p + "input_layernorm.weight", self.ln1[i].data
            yield p + "post_attention_layernorm.weight", self.ln2[i].data
        yield "model.norm.weight", self.norm.data
        yield "lm_head.weight", self.embed.data[: c.vocab_size]

    def state_dict(self, *a, **k):
        """HF parameter names; with LoRA enabled also the adapters under peft's names (`lora_named`)."""
        self.sync_optimizer()
        if self.lora is None:
            return {n: t.clone() for n, t in self._hf_named()}
        # peft's PeftModel.state_dict(): everything under `base_model.model.`, the wrapped projections' own weights under
        # `.base_layer.` -- the names a reference run with --peft writes and reads back (ecg_byte/main.py:153-155,193-195)
        def peft_name(n):
            if n.endswith("_proj.weight"):
                n = n[: -len("weight")] + "base_layer.weight"
            return "base_model.model." + n
        sd = {peft_name(n): t.clone() for n, t in self._hf_named()}
        sd.update({n: t.clone() for n, t in self.lora_named()})
        return sd

    def load_state_dict(self, sd, strict=True):
        """Accepts HF names, and a peft checkpoint's spelling of them (`base_model.model.` prefix, `.base_layer.` infix)."""
        self.sync_optimizer()
        def canon(n):
            if "lora_" in n:
                return n
            return n.replace("base_model.model.", "", 1).replace(".base_layer.", ".") if n.startswith("base_model.model.") else n
        sd = {canon(n): t for n, t in sd.items()}
        names = dict(self._hf_named())
        if self.lora is not None:
            names.update(dict(self.lora_named()))
        missing = [n for n in names if n not in sd and n != "lm_head.weight"]
        unexpected = [n for n in sd if n not in names and not n.endswith("rotary_emb.inv_freq")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for n, t in names.items():
                if n in sd and n != "lm_head.weight":
                    t.copy_(sd[n].to(device=t.device, dtype=t.dtype))
        self._t.clear()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    # ---- helpers --------------------------------------------------------------------------
    def _shadow(self, key, p):
        """Transposed copy of a weight.  The optimizer drops the copies of the parameters it updated (`_drop_shadows`): the frozen
        base of a LoRA run is transposed once, not once per step."""
        ver = p._version
        hit = self._t.get(key)
        if hit is None or self._t_version.get(key) != ver:
            hit = self._t[key] = (ops.transpose(p.data), id(p))
            self._t_version[key] = ver
        return hit[0]

    def _refresh_lora_shadows(self):
        """The transposed copies of every adapter (A^T for lora_dx, B^T for dt = dy B) in ONE launch: 2 per site and layer, a few KB each --
        as 224 single transposes per step they were launch-bound (1.5 ms)."""
        sites = [(i, k, s) for i, layer in enumerate(self.lora) for k, s in layer.items()]
        tensors = [t for _, _, s in sites for t in (s.A.data, s.B.data)]
        plan = getattr(self, "_lora_tplan", None)
        if plan is None or not plan.valid_for(tensors):
            plan = self._lora_tplan = ops.TransposePlan(tensors)
        if all(self._t_version.get(("lora_At", id(s))) == s.A._version and ("lora_At", id(s)) in self._t and
               self._t_version.get(("lora_Bt", id(s))) == s.B._version and ("lora_Bt", id(s)) in self._t for _, _, s in sites):
            return                                                       # nothing was updated since the copies were made
        dst = ops.transpose_multi(plan)
        for n, (_, _, s) in enumerate(sites):
            self._t[("lora_At", id(s))] = (dst[2 * n], id(s.A)); self._t_version[("lora_At", id(s))] = s.A._version
            self._t[("lora_Bt", id(s))] = (dst[2 * n + 1], id(s.B)); self._t_version[("lora_Bt", id(s))] = s.B._version

    def _drop_shadows(self, param_ids):
        for key in [k for k, (_, pid) in self._t.items() if pid in param_ids]:
            del self._t[key]

    def _dx(self, dy, key, p):
        """dx = dy . W for a weight stored [out, in] (the input gradient of y = x W^T): the NN kernel reads W as it lies (K-sliced when the output has
        few tiles and the contraction is long); shapes below its 256x256 tile go through the NT kernel on a transposed shadow copy (`_shadow`)."""
        if ops.nn_eligible(dy.shape[0], p.shape[1], p.shape[0]):
            return ops.gemm_nn(dy, p.data)
        splits = ops.nn_splitk_plan(dy.shape[0], p.shape[1], p.shape[0])
        if splits:                                                       # few tiles, a long contraction (the loss head: dlogits . E over the vocabulary)
            return ops.gemm_nn_splitk(dy, p.data, splits)
        return ops.gemm_nt(dy, self._shadow(key, p))

    def _proj(self, i, key, x, w, training=False, keep=False, rope=None, t_pre=None, sync=None):
        """One projection of layer i ("qkv", "o", "gu", "down"): x W^T, plus the LoRA branch of the site when adapters are on
        (in the same launch).  rope = (cos, sin, columns): the q|k|v projection leaves with its q and k heads rotated (apply_rotary_pos_emb in the GEMM's
        epilogue where the kernel takes the shape, else as the separate pass).  Returns (y, what the adapter's backward needs or None)."""
        if self.lora is None:
            if rope is not None:
                return ops.gemm_nt_rope(x, w, rope[0], rope[1], rope[2]), None
            return ops.gemm_nt(x, w), None
        return self.lora[i][key].project(x, w, training, keep, rope=rope, t_pre=t_pre, sync=sync)

    def _proj_glu(self, i, x, training=False, keep=False, keep_gu=True, t_pre=None):
        """The MLP's gate|up projection with the GLU in the GEMM's epilogue: returns (gate|up [T, 2I] or None, act(gate) * up [T, I],
        adapter state).  gate|up is what the backward needs; inference passes keep_gu=False and the tensor is never written."""
        w = self.wgu[i].data
        if self.lora is not None:
            (gu, hm), ls = self.lora[i]["gu"].project(x, w, training, keep, glu=(self.gemma, keep_gu), t_pre=t_pre)
            return gu, hm, ls
        if ops.glu_fusable(x.shape[0], w.shape[0] // 2):
            gu, hm = ops.gemm_nt_glu(x, w, gelu_tanh=self.gemma, keep_gu=keep_gu)
        else:
            gu = ops.gemm_nt(x, w)
            hm = ops.glu_fwd(gu, gelu_tanh=self.gemma)
        return gu, hm, None

    def _rope_tables(self, position_ids):
        pos = position_ids.reshape(-1).float()
        fr = pos[:, None] * self.inv_freq[None, :].float()
        return fr.cos().contiguous(), fr.sin().contiguous()

    def _attn_materialised(self, qkv, mask, B, S, drop=None):
        """Attention through the [B*Hq, S, S] score tensor (head-batched GEMMs + softmax kernel): any head_dim that is a
        multiple of 64 (Gemma: 256); S must be a multiple of 64 (GEMM K-step).  Returns (output [B*S, Hq*D], probabilities).
        drop = (p, seed): dropout on the probabilities before P.V (GPT-2's attn_dropout, modeling_gpt2.py:214); then the second
        return value is the pair (P, dropped P) the backward needs."""
        c = self.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        G, QKV, dev = Hq // Hkv, self.qkv, qkv.device
        assert S % 64 == 0, "materialised attention needs a sequence length that is a multiple of 64"
        P = torch.empty((B * Hq, S, S), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt_heads((qkv, 0), QKV, (qkv, Hq * D), QKV, P, S, S, S, D, 1.0, B * Hq, Hq,
                          S * QKV, D, 1, S * QKV, D, G, Hq * S * S, S * S)
        ops.softmax_causal_fwd_(P, mask, Hq, 1.0 / math.sqrt(D))
        vT = torch.empty((B, Hkv, D, S), dtype=torch.bfloat16, device=dev)
        ops.transpose_strided(qkv, Hq * D + Hkv * D, vT, 0, S, D, QKV, S, B * Hkv, Hkv, S * QKV, D, Hkv * D * S, D * S)
        ao = torch.empty((B * S, Hq * D), dtype=torch.bfloat16, device=dev)
        Pd = ops.dropout(P, drop[0], drop[1]) if drop else P
        ops.gemm_nt_heads(Pd, S, vT, S, ao, Hq * D, S, D, S, 1.0, B * Hq, Hq, Hq * S * S, S * S, 1,
                          Hkv * D * S, D * S, G, S * Hq * D, D)
        return ao, ((P, Pd) if drop else P)

    def _attn_materialised_bwd(self, qkv, d_ao, P, B, S, drop=None):
        """d_qkv of `_attn_materialised` (same layout as qkv).  P: its second return value."""
        c = self.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        G, QKV, dev = Hq // Hkv, self.qkv, qkv.device
        T, scale = B * S, 1.0 / math.sqrt(D)
        P, Pd = P if drop else (P, P)
        d_qkv = torch.empty((T, QKV), dtype=torch.bfloat16, device=dev)
        dP = torch.empty((B * Hq, S, S), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt_heads(d_ao, Hq * D, (qkv, Hq * D + Hkv * D), QKV, dP, S, S, S, D, 1.0, B * Hq, Hq,
                          S * Hq * D, D, 1, S * QKV, D, G, Hq * S * S, S * S)   # dP = dO . V^T
        if drop:
            ops.dropout(dP, drop[0], drop[1], out=dP)                            # the same mask, on the gradient
        ops.softmax_bwd_(P, dP, scale)                                           # dP <- dS
        kT = torch.empty((B, Hkv, D, S), dtype=torch.bfloat16, device=dev)
        ops.transpose_strided(qkv, Hq * D, kT, 0, S, D, QKV, S, B * Hkv, Hkv, S * QKV, D, Hkv * D * S, D * S)
        ops.gemm_nt_heads(dP, S, kT, S, (d_qkv, 0), QKV, S, D, S, 1.0, B * Hq, Hq, Hq * S * S, S * S, 1,
                          Hkv * D * S, D * S, G, S * QKV, D)                     # dQ = dS . K
        tmpT = torch.empty((B * Hq, S, S), dtype=torch.bfloat16, device=dev)
        qT = torch.empty((B, Hq, D, S), dtype=torch.bfloat16, device=dev)
        ops.transpose_strided(qkv, 0, qT, 0, S, D, QKV, S, B * Hq, Hq, S * QKV, D, Hq * D * S, D * S)
        doT = torch.empty((B, Hq, D, S), dtype=torch.bfloat16, device=dev)
        ops.transpose_strided(d_ao, 0, doT, 0, S, D, Hq * D, S, B * Hq, Hq, S * Hq * D, D, Hq * D * S, D * S)
        dkv32 = torch.zeros((2, B, S, Hkv * D), dtype=torch.float32, device=dev)
        for which, (src, rhs) in enumerate(((dP, qT), (Pd, doT))):                # dK = dS^T . Q ; dV = P^T . dO
            ops.transpose_strided(src, 0, tmpT, 0, S, S, S, S, B * Hq, 1, S * S, 0, S * S, 0)
            for j in range(G):   # the G query heads of a KV head accumulate into the same fp32 tile
                ops.gemm_nt_heads((tmpT, j * S * S), S, (rhs, j * D * S), S, dkv32[which], Hkv * D, S, D, S, 1.0,
                                  B * Hkv, Hkv, Hq * S * S, G * S * S, 1, Hq * D * S, G * D * S, 1,
                                  S * Hkv * D, D, accumulate_f32=True)
        d_qkv[:, Hq * D: Hq * D + Hkv * D] = dkv32[0].view(T, Hkv * D).to(torch.bfloat16)
        d_qkv[:, Hq * D + Hkv * D:] = dkv32[1].view(T, Hkv * D).to(torch.bfloat16)
        return d_qkv

    # ---- gradient storage ---------------------------------------------------------------------
    # ---- optimizer step overlapped with the next forward (HipAdam(overlap=True)) ---------------------------------------------------
    def _opt_groups(self):
        """Trainable parameters in the order the FORWARD pass first reads them: ("embed", ...), (0, layer 0), ..., ("norm", ...)."""
        if self.lora is not None:
            return [(i, list(self.lora[i].parameters())) for i in range(self.cfg.num_hidden_layers)]
        L = self.cfg.num_hidden_layers
        return ([("embed", [self.embed])] +
                [(i, [self.ln1[i], self.wqkv[i], self.wo[i], self.ln2[i], self.wgu[i], self.wdown[i]]) for i in range(L)] +
                [("norm", [self.norm])])

    def _wait_group(self, key):
        """The current stream waits until the optimizer has updated parameter group `key` (no-op without an overlapped step in flight)."""
        ev = self._opt_ready.pop(key, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def sync_optimizer(self):
        """The current stream waits for the whole optimizer step (everything that reads or writes parameters or gradients outside the training
        forward calls this first)."""
        if self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
            self._opt_done = None
            self._opt_ready.clear()

    def _grad_layout(self):
        """Every trainable gradient lives in ONE flat bf16 buffer, laid out in the order backward finishes them: layer L-1's
        tensors first, layer 0's last, then the embedding table and the final norm (full fine-tune) -- so "the gradients of
        layers i..j are final" is one contiguous range, which parallel.GradAllReduce sends as one all-reduce per >= 25 MB
        (the reference's DDP buckets, ecg_byte/main.py:165) instead of one per tensor, and the weight-gradient GEMMs write
        straight into it.  `param.grad` is a view of its slice.  Slices start on 256-byte boundaries; the gaps stay zero."""
        frozen = self.lora is not None
        key = (frozen, self.v_pad, id(self.lora))
        if self._gflat is not None and self._gflat_key == key:
            return
        groups = self._grad_groups(frozen)
        off, plan, ranges = 0, [], []
        for ps in groups:
            lo = off
            for p in ps:
                plan.append((p, off))
                off += (p.numel() + 127) // 128 * 128
            ranges.append((lo, off))
        self._gflat = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self._gview = {id(p): self._gflat[o: o + p.numel()].view(p.shape) for p, o in plan}
        self._granges = ranges          # [L - 1 - i] = layer i's range; [L] = embed + final norm
        self._gflat_key = key

    def _grad_groups(self, frozen):
        """Trainable parameters in the order backward finishes their gradients: one group per layer, last layer first, then the rest."""
        L = self.cfg.num_hidden_layers
        groups = []
        for i in reversed(range(L)):
            groups.append(list(self.lora[i].parameters()) if frozen else
                          [self.wqkv[i], self.wo[i], self.wgu[i], self.wdown[i], self.ln1[i], self.ln2[i]])
        if not frozen:
            groups.append([self.embed, self.norm])
        return groups

    def _grad_slot(self, param):
        """(view, accumulate): the parameter's slice of the flat buffer, and whether it already holds a gradient this
        backward must add to (gradient accumulation: `param.grad` survived since the last zero_grad)."""
        view = self._gview[id(param)]
        if param.grad is None:
            return view, False
        if param.grad.data_ptr() != view.data_ptr():
            view.copy_(param.grad)
        return view, True

    def _wgrad(self, param, dy, xin, alpha=1.0):
        """param.grad (+)= alpha * dy^T . xin, written by the TN GEMM into the flat buffer."""
        self._wgrad_rows(param, [(0, param.shape[0], dy, xin)], alpha)

    def _wgrad_rows(self, param, parts, alpha=1.0):
        """parts = [(lo, hi, dy, xin)]: rows [lo, hi) of param.grad (+)= alpha * dy^T . xin (the stacked LoRA adapters: one product per
        block; rows no part covers keep the zeros the flat buffer was created with)."""
        view, acc = self._grad_slot(param)
        for lo, hi, dy, xin in parts:
            dst = view[lo:hi]
            ops.gemm_tn(dy, xin, alpha=alpha, out=dst, accumulate=acc)
        param.grad = view

    def _lora_agrad(self, param, x, dt, n_sub, n_fields, scale, p, seed):
        """Rows [0, 16 n_sub) of a stacked LoRA A's gradient (+)= scale / (1 - p) * dt^T . (mask . x) (ecgb_lora_da); the rows past them keep
        the zeros the flat buffer was created with."""
        view, acc = self._grad_slot(param)
        ops.lora_da(x, dt, view[: 16 * n_sub], n_sub, n_fields, scale, p, seed, accumulate=acc)
        param.grad = view

    def _vgrad(self, param, g):
        """param.grad (+)= g (any float dtype, converted to bf16)."""
        view, acc = self._grad_slot(param)
        if acc:
            ops.add(view, g.to(torch.bfloat16), out=view)
        else:
            view.copy_(g)
        param.grad = view

    # ---- forward ------------------------------------------------------------------------------
    def _forward_loss(self, input_ids, attention_mask, labels, position_ids):
        c = self.cfg
        H, I, D, Hq, Hkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        G = Hq // Hkv
        dev = self.device
        ops.require_current(dev)
        input_ids = input_ids.to(dev).contiguous()
        B, S = input_ids.shape
        mask = (attention_mask.to(dev).float() if attention_mask is not None else torch.ones(B, S, device=dev)).contiguous()
        if position_ids is None:
            position_ids = torch.arange(S, device=dev)[None].expand(B, S)
        position_ids = position_ids.to(dev)
        labels = labels.to(dev)
        if S % 64:   # the GEMM K-step is 64 rows/columns: masked, unlabelled positions on the LEFT change neither the loss nor any
            lpad = 64 - S % 64   # gradient (the reference's default --pad_to_max 1000 gives rows of 1004, data_loader.py:123)
            input_ids = torch.cat([torch.zeros((B, lpad), dtype=input_ids.dtype, device=dev), input_ids], 1).contiguous()
            mask = torch.cat([torch.zeros((B, lpad), device=dev), mask], 1).contiguous()
            position_ids = torch.cat([torch.zeros((B, lpad), dtype=position_ids.dtype, device=dev), position_ids], 1)
            labels = torch.cat([torch.full((B, lpad), -100, dtype=labels.dtype, device=dev), labels], 1)
            S += lpad
        T = B * S
        cos, sin = self._rope_tables(position_ids)
        rope = (cos, sin, (Hq + Hkv) * D)                                 # what the q|k|v projection rotates on its way out
        QKV = self.qkv
        scale = 1.0 / math.sqrt(D)
        saved = []
        self._wait_group("embed")                                    # (an overlapped optimizer step: this group's update has landed)
        x = ops.embed_fwd(input_ids.view(-1), self.embed.data, self.embed_scale)          # [T, H]
        delta = None
        for i in range(c.num_hidden_layers):
            self._wait_group(i)
            h1, rstd1, x1 = ops.rmsnorm_fwd(x, self.ln1[i].data, c.rms_norm_eps, residual=delta, gemma=self.gemma)
            ls = [None] * 4
            qkv, ls[0] = self._proj(i, "qkv", h1, self.wqkv[i].data, self.training, rope=rope)    # [T, QKV], q and k heads rotated
            if self.fused_attention:
                ao, P = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)       # P slot holds the row log-sum-exps
            else:
                ao, P = self._attn_materialised(qkv, mask, B, S)
            attn_delta, ls[1] = self._proj(i, "o", ao, self.wo[i].data, self.training)   # [T, H]
            h2, rstd2, x2 = ops.rmsnorm_fwd(x1, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma)
            gu, hm, ls[2] = self._proj_glu(i, h2, self.training)                         # [T, 2I], [T, I]
            delta, ls[3] = self._proj(i, "down", hm, self.wdown[i].data, self.training)
            saved.append((x1, rstd1, h1, qkv, P, ao, x2, rstd2, h2, gu, hm, ls))
            x = x2
        self._wait_group("norm")
        hf, rstdf, xf = ops.rmsnorm_fwd(x, self.norm.data, c.rms_norm_eps, residual=delta, gemma=self.gemma)

        loss, dhf = self._loss_head(hf, labels, B, S)
        self._saved = (saved, input_ids, mask, cos, sin, (xf, rstdf), dhf, (B, S))
        return loss.squeeze(0)

    def _loss_head(self, hf, labels, B, S):
        """Tied lm_head + ForCausalLMLoss (loss_utils.py:24-47) over the final hidden states hf [B*S, H]: returns (loss, d loss / d hf)
        and leaves d loss / d E of the head in self.embed_grad_head (bf16 [v_pad, H]: the table's slice of the flat gradient buffer, or a buffer of
        its own when that slice holds a kept gradient or no gradient is being taken; unless the base is frozen)."""
        c = self.cfg
        H = c.hidden_size
        dev = self.device
        T = B * S
        # ---- loss head: ForCausalLMLoss shifts (loss_utils.py:39-41): row t predicts labels[t+1]
        shifted = torch.full((B, S), -100, dtype=torch.int64, device=dev)
        shifted[:, :-1] = labels[:, 1:]
        shifted = shifted.view(-1)
        self._unlabelled_rows = shifted == -100                          # (backward: a row takes no gradient only if it is masked AND carries no loss)
        inv_count = ops.count_labels(shifted, c.vocab_size)
        rows = torch.arange(T, device=dev) if self.full_logits else torch.nonzero(shifted != -100).view(-1)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        frozen = self.lora is not None
        if not frozen:
            # The head's share of dE goes straight into the table's slice of the flat gradient buffer when that slice is free (training, no gradient kept from an
            # earlier backward): the backward then finds it in place -- no [v_pad, H] buffer of its own (0.54 GB at Llama-3.2-1B's vocabulary) and no copy of it.
            # (`_wait_group("embed")` above must precede this: with the overlapped optimizer the update of the table may still be reading its gradient slice.)
            if torch.is_grad_enabled() and self.embed.grad is None:
                self._grad_layout()
                self.embed_grad_head = self._gview[id(self.embed)]
            else:
                if getattr(self, "_head_buf", None) is None:
                    self._head_buf = torch.empty((self.v_pad, H), dtype=torch.bfloat16, device=dev)
                self.embed_grad_head = self._head_buf
        dhf = torch.zeros((T, H), dtype=torch.bfloat16, device=dev)

        chunk = 4096
        first = True
        for s0 in range(0, rows.numel(), chunk):
            r = rows[s0:s0 + chunk]
            n = r.numel()
            npad = (n + 255) // 256 * 256                                # whole 256-row tiles: the head's two products then run on the four-wave kernel (0.75 of the MFMA pipe
                                                                         # against 0.37 on the eight-wave one: profiles/r05/train_pmc.json); the rows are the weight-gradient product's K (step 64)
            hr = torch.zeros((npad, H), dtype=torch.bfloat16, device=dev)     # rows past n: zero states, label -100 -> zero logits, zero dlogits
            hr[:n] = hf.index_select(0, r)                               # gather (plumbing)
            lab = torch.full((npad,), -100, dtype=torch.int64, device=dev)
            lab[:n] = shifted.index_select(0, r)
            logits = ops.gemm_nt(hr, self.embed.data)                    # [npad, v_pad]
            ops.ce_fwd_bwd_(logits, lab, inv_count, loss, c.vocab_size)  # logits <- dlogits
            dhr = self._dx(logits, "embed", self.embed)                  # [npad, H] = dlogits . E
            dhf.index_copy_(0, r, dhr[:n])
            if frozen:
                continue                                                 # frozen base: no embedding / lm_head gradient
            ops.gemm_tn(logits, hr, out=self.embed_grad_head, accumulate=not first)    # dE (+)= dlogits^T . h, straight from the row-major operands
            first = False
        if not frozen and first:                                         # no labelled row at all
            self.embed_grad_head.zero_()
        return loss, dhf

    def _embedding_grad(self, param, head, ids, g, scale, skip_id):
        """param.grad (+)= head (the tied lm_head's share, or None) + the scatter of the rows of g by ids -- in the flat gradient buffer,
        without atomics (ecgb_embed_bwd_sorted)."""
        view, acc = self._grad_slot(param)
        if head is not None:
            if head.data_ptr() == view.data_ptr():
                pass                                                     # (the loss head wrote it here: _loss_head)
            elif acc:
                ops.add(view, head, out=view)
            else:
                view.copy_(head)
        elif not acc:
            view.zero_()
        ops.embed_bwd_sorted_(ids, g, view, scale, skip_id)
        param.grad = view

    def _embedding_grad_sparse(self, live_ids, g, skip_id):
        """Data-parallel form of _embedding_grad: the table's slice of the flat buffer already holds the MEAN over the ranks of the tied head's dE (its
        all-reduce left when backward started and is waited for here); every rank reduces the rows of g by id into a compact [distinct ids, H] table
        (ecgb_embed_bwd_sorted on the ranks of the ids), the ranks exchange (ids, rows) (GradAllReduce.exchange_rows) and each adds all of them, divided
        by the world size, in rank order: avg(head) + sum_r scatter_r / W = avg(head + scatter), the same gradient the dense exchange produces (up to
        one more bf16 rounding of the sparse rows), with a tail of a few MB after the last layer instead of the 0.54 GB table."""
        gs = self.grad_sync
        view = self._gview[id(self.embed)]
        ids = live_ids if skip_id < 0 else live_ids.masked_fill(live_ids == skip_id, -1)
        uniq, inv = torch.unique(ids, sorted=True, return_inverse=True)
        if uniq.numel() and int(uniq[0]) < 0:                                 # -1: masked / padding rows, no gradient
            uniq, inv = uniq[1:], inv - 1
        rows = torch.zeros((max(int(uniq.numel()), 1), view.shape[1]), dtype=torch.bfloat16, device=view.device)
        ops.embed_bwd_sorted_(inv, g, rows, self.embed_scale, -1)             # compact: row j = the sum over the tokens whose id is uniq[j]
        parts = gs.exchange_rows(uniq, rows[: uniq.numel()])
        v_lo, v_hi = view.data_ptr(), view.data_ptr() + view.numel() * 2
        overlaps = lambda t: t.data_ptr() < v_hi and t.data_ptr() + t.numel() * t.element_size() > v_lo
        for t, work, divide in gs.pending:                                    # the dense part (issued first) has landed
            if overlaps(t):
                work.wait()
                if divide and gs.world > 1:
                    t.div_(gs.world)
        gs.pending = [(t, w, dv) for t, w, dv in gs.pending if not overlaps(t)]
        W = float(max(1, len(parts)))
        for ids_r, rows_r in parts:                                           # rank order: the same bits on every rank
            if ids_r.numel():
                ops.embed_bwd_sorted_(ids_r, rows_r, view, 1.0 / W, -1)

    # ---- inference -----------------------------------------------------------------------------
    def _hidden_states(self, input_ids, attention_mask=None, position_ids=None, kv_out=None):
        """Final-normed hidden states [B*S, H] of a whole (left-padded) batch, nothing saved for backward.  kv_out: optional
        list of per-layer caches [B, cap, 2*Hkv*D]; rows [:S] receive the roped keys and the values (DynamicCache.update,
        cache_utils.py:408-470)."""
        self.sync_optimizer()
        c = self.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        dev = self.device
        ops.require_current(dev)
        input_ids = input_ids.to(dev).contiguous()
        B, S = input_ids.shape
        mask = (attention_mask.to(dev).float() if attention_mask is not None else torch.ones(B, S, device=dev)).contiguous()
        if position_ids is None:
            position_ids = torch.arange(S, device=dev)[None].expand(B, S)
        position_ids = position_ids.to(dev)
        lpad = 0
        if not self.fused_attention and S % 64:   # the materialised path needs S % 64 == 0: masked positions on the LEFT change nothing
            lpad = 64 - S % 64
            input_ids = torch.cat([torch.zeros((B, lpad), dtype=input_ids.dtype, device=dev), input_ids], 1).contiguous()
            mask = torch.cat([torch.zeros((B, lpad), device=dev), mask], 1).contiguous()
            position_ids = torch.cat([torch.zeros((B, lpad), dtype=position_ids.dtype, device=dev), position_ids], 1)
            S_out, S = S, S + lpad
        cos, sin = self._rope_tables(position_ids)
        rope = (cos, sin, (Hq + Hkv) * D)                                 # what the q|k|v projection rotates on its way out
        QKV = self.qkv
        scale = 1.0 / math.sqrt(D)
        x = ops.embed_fwd(input_ids.view(-1), self.embed.data, self.embed_scale)
        delta = None
        for i in range(c.num_hidden_layers):
            h1, _, x = ops.rmsnorm_fwd(x, self.ln1[i].data, c.rms_norm_eps, residual=delta, gemma=self.gemma)
            qkv, _ = self._proj(i, "qkv", h1, self.wqkv[i].data, rope=rope)
            if kv_out is not None:
                kv_out[i][:, :S - lpad].copy_(qkv.view(B, S, QKV)[:, lpad:, Hq * D:])
            if self.fused_attention:
                ao, _ = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
            else:
                ao, _ = self._attn_materialised(qkv, mask, B, S)
            attn_delta, _ = self._proj(i, "o", ao, self.wo[i].data)
            if self.lora is not None and not self.training:
                site = self.lora[i]["gu"]
                h2, _, x, t2 = ops.rmsnorm_fwd(x, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma, lora=(site.A.data, site.scale))
            else:
                (h2, _, x), t2 = ops.rmsnorm_fwd(x, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma), None
            _, hm, _ = self._proj_glu(i, h2, keep_gu=False, t_pre=t2)
            delta, _ = self._proj(i, "down", hm, self.wdown[i].data)
        hf, _, _ = ops.rmsnorm_fwd(x, self.norm.data, c.rms_norm_eps, residual=delta, gemma=self.gemma)
        if lpad:
            hf = hf.view(B, S, -1)[:, lpad:].reshape(B * S_out, -1)
        return hf

    def _eval_loss(self, input_ids, attention_mask, labels, position_ids):
        """ForCausalLMLoss (loss_utils.py:24-47) without a backward: forward-only hidden states, lm_head + cross entropy
        over the rows whose shifted label is not -100."""
        c = self.cfg
        dev = self.device
        B, S = input_ids.shape
        hf = self._hidden_states(input_ids, attention_mask, position_ids)
        labels = labels.to(dev)
        shifted = torch.full((B, S), -100, dtype=torch.int64, device=dev)
        shifted[:, :-1] = labels[:, 1:]
        shifted = shifted.view(-1)
        inv_count = ops.count_labels(shifted, c.vocab_size)
        rows = torch.nonzero(shifted != -100).view(-1)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        for s0 in range(0, rows.numel(), 4096):
            r = rows[s0:s0 + 4096]
            logits = ops.gemm_nt(hf.index_select(0, r), self.embed.data)
            ops.ce_fwd_bwd_(logits, shifted.index_select(0, r), inv_count, loss, c.vocab_size)
        return loss.squeeze(0)

    def _decode_step(self, tokens, pos, mask, caches, n, n_dev=None, scratch=None, scratch_one=None, epoch_advanced=False):
        """Hidden state [B, H] of one new token per sequence, written at cache row n-1 (n = keys valid after the update).
        n_dev: int32[1] device tensor holding n -- then nothing in the launches depends on the step (graph replay); scratch: the
        split decode attention's buffer the captured step owns (scratch_one: the one-launch attention's).  epoch_advanced: the caller advances the model's step counter
        itself (ops.decode_advance_(epoch=...) behind the step, as the captured loop does); else the step does it here, a launch of its own."""
        c = self.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        QKV = self.qkv
        scale = 1.0 / math.sqrt(D)
        if pos.dtype == torch.int64 and self.inv_freq.dtype == torch.float32:
            cos, sin = ops.rope_table(pos, self.inv_freq)               # (the four element-wise kernels of _rope_tables in one launch, the same bits)
        else:
            cos, sin = self._rope_tables(pos)
        x = ops.embed_fwd(tokens, self.embed.data, self.embed_scale)    # [B, H]
        delta = None
        # round 6: the o and down sites of a step with adapters form t = scale * x A^T inside the projection's launch (two launches a layer less), the launch's workgroups meeting
        # on flags that hold the model's step counter -- a device word with a new value every step
        sync = None
        if self.lora is not None and not self.training and self.decode_lora_one:
            sync = self._decode_epoch(x.device)
            if not epoch_advanced:
                sync.add_(1)
        # (opt-in: the fused kernels of csrc/decode.hip are bit-exact but measured no faster than the separate ones at the C5 shape -- 2.20 against 2.14 ms a token; every
        #  kernel of the step sits on the ~5 us a dependent launch of a replayed graph costs, whatever it fuses: profiles/r05/README.md)
        if getattr(self, "decode_fused", False) and not self.training and ops.decode_fusable(x.shape[0], c.hidden_size, Hq, Hkv, D):
            return self._decode_layers_fused(x, cos, sin, mask, caches, n if n_dev is None else n_dev)
        for i in range(c.num_hidden_layers):
            if self.lora is not None and not self.training:             # adapters: the norm kernel forms the site's t = scale * h A^T on its way out
                site = self.lora[i]["qkv"]
                h1, _, x, t1 = ops.rmsnorm_fwd(x, self.ln1[i].data, c.rms_norm_eps, residual=delta, gemma=self.gemma, lora=(site.A.data, site.scale))
            else:
                (h1, _, x), t1 = ops.rmsnorm_fwd(x, self.ln1[i].data, c.rms_norm_eps, residual=delta, gemma=self.gemma), None
            qkv, _ = self._proj(i, "qkv", h1, self.wqkv[i].data, t_pre=t1)         # [B, QKV]
            ns = ops.decode_splits(caches[i].shape[1], qkv.shape[0], Hq)
            nn = n if n_dev is None else n_dev
            if self.decode_attn_one and ops.decode_one_ok(qkv.shape[0], Hq, D, ns, caches[i].shape[1]):
                # round 6: RoPE + cache append + scores + values + combine in ONE launch (the bits of the four: tests/test_gpu_decode_fused.py)
                ao = ops.attn_decode_one(qkv, cos, sin, caches[i], mask, nn, Hq, Hkv, D, scale, ns, scratch=scratch_one if n_dev is not None else None)
            else:
                ops.rope_append_(qkv, cos, sin, Hq, Hkv, D, caches[i], nn)   # RoPE on q and k, the rotated k and v into the cache: one launch
                if ns > 1:
                    ao = ops.attn_decode_split(qkv, caches[i], mask, nn, Hq, Hkv, D, scale, ns, scratch=scratch if n_dev is not None else None)
                elif n_dev is None:
                    ao = ops.attn_decode(qkv, caches[i], mask, n, Hq, Hkv, D, scale)
                else:
                    ao = ops.attn_decode_dyn(qkv, caches[i], mask, n_dev, Hq, Hkv, D, scale)
            attn_delta, _ = self._proj(i, "o", ao, self.wo[i].data, sync=sync)
            if self.lora is not None and not self.training:             # (the gate|up site's t on the norm's way out too: one launch less a layer, the same bits)
                site = self.lora[i]["gu"]
                h2, _, x, t2 = ops.rmsnorm_fwd(x, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma, lora=(site.A.data, site.scale))
            else:
                (h2, _, x), t2 = ops.rmsnorm_fwd(x, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma), None
            _, hm, _ = self._proj_glu(i, h2, keep_gu=False, t_pre=t2)
            delta, _ = self._proj(i, "down", hm, self.wdown[i].data, sync=sync)
        hf, _, _ = ops.rmsnorm_fwd(x, self.norm.data, c.rms_norm_eps, residual=delta, gemma=self.gemma)
        return hf

    def _decode_epoch(self, dev):
        """The decode steps' counter (int32[1] on the device, 1 at first, + 1 every step for the model's lifetime)."""
        e = self.__dict__.get("_dec_epoch")
        if e is None or e.device != dev:
            e = self.__dict__["_dec_epoch"] = torch.ones(1, dtype=torch.int32, device=dev)
        return e

    def _decode_layers_fused(self, x, cos, sin, mask, caches, n):
        """The layers of a decode step for one or two sequences on the fused kernels of csrc/decode.hip: per layer q|k|v with its norm and adapter branch in one launch,
        RoPE + cache append + attention in two, o with its adapter's down-projection in one, the MLP in two to four (round 4: twelve launches a layer).  n: keys valid after
        the append, an int or the graph's int32[1] device tensor.  The same bits as the separate kernels (tests/test_gpu_decode_fused.py)."""
        c = self.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        B, cap = x.shape[0], caches[0].shape[1]
        scale = 1.0 / math.sqrt(D)
        # attention: round 4's kernels (RoPE + append, scores, values, combine) unless `decode_fused_attn` is set -- the two-launch form of csrc/decode.hip is bit-exact
        # and reads a KV group's keys once, but measured SLOWER at the C5 shape (82 against 25 us a layer: profiles/r05/README.md); kept for the record and the tests
        fused_attn = bool(getattr(self, "decode_fused_attn", False))
        dyn = torch.is_tensor(n)
        ns, ns4 = ops.decode_attn_splits(cap), ops.decode_splits(cap, B, Hq)
        pool = self.__dict__.setdefault("_dec_scratch", {})
        key = (B, cap, ns, ns4, fused_attn)
        scratch = scratch4 = None
        if key not in pool:                                              # (first use: the warm-up step in front of a capture, never inside one)
            if len(pool) >= 4:
                pool.pop(next(iter(pool)))
            pool[key] = (ops.decode_attn_scratch(cap, B, Hq, Hkv, D, ns, x.device) if fused_attn else None,
                         ops.decode_split_scratch(cap, B, Hq, D, ns4, x.device) if (ns4 > 1 and not fused_attn) else None)
        scratch, scratch4 = pool[key]
        glu = 2 if self.gemma else 1
        delta = None
        for i in range(c.num_hidden_layers):
            L = self.lora[i] if self.lora is not None else None
            site = (lambda k: (L[k].A.data, 16 * L[k].n_sub, L[k].scale, L[k].B.data)) if L is not None else (lambda k: None)
            if L is None:
                qkv, x = ops.decode_norm_gemv(x, delta, self.ln1[i].data, c.rms_norm_eps, self.gemma, self.wqkv[i].data)
            else:   # (with adapters: the norm + LoRA-down kernel, then the projection with t given -- redone by each of the projection's workgroups the 48 rows of A cost more than the launch)
                h1, _, x, t1 = ops.rmsnorm_fwd(x, self.ln1[i].data, c.rms_norm_eps, residual=delta, gemma=self.gemma, lora=(L["qkv"].A.data, L["qkv"].scale))
                qkv = ops.decode_gemv(h1, self.wqkv[i].data, lora=site("qkv"), t=t1)
            if fused_attn:
                ao = ops.decode_attn(qkv, cos, sin, caches[i], mask, n, Hq, Hkv, D, scale, ns, scratch)
            else:   # RoPE + append, then the attention kernels of round 4 (one workgroup a head for short caches, split over workgroups for long ones)
                ops.rope_append_(qkv, cos, sin, Hq, Hkv, D, caches[i], n)
                if ns4 > 1:
                    ao = ops.attn_decode_split(qkv, caches[i], mask, n, Hq, Hkv, D, scale, ns4, scratch=scratch4 if dyn else None)
                elif dyn:
                    ao = ops.attn_decode_dyn(qkv, caches[i], mask, n, Hq, Hkv, D, scale)
                else:
                    ao = ops.attn_decode(qkv, caches[i], mask, n, Hq, Hkv, D, scale)
            attn_delta = ops.decode_gemv(ao, self.wo[i].data, lora=site("o"))
            if L is None:
                hm, x = ops.decode_norm_gemv(x, attn_delta, self.ln2[i].data, c.rms_norm_eps, self.gemma, self.wgu[i].data, glu=glu)
                delta = ops.decode_gemv(hm, self.wdown[i].data)
            else:
                # (with adapters the gate|up site keeps the norm + LoRA-down kernel: redone by each of the projection's 2 000 workgroups, A's 32 rows would be read 2 000 times)
                h2, _, x, t2 = ops.rmsnorm_fwd(x, self.ln2[i].data, c.rms_norm_eps, residual=attn_delta, gemma=self.gemma, lora=(L["gu"].A.data, L["gu"].scale))
                _, hm, _ = self._proj_glu(i, h2, keep_gu=False, t_pre=t2)
                sd = L["down"]
                delta = ops.decode_gemv(hm, self.wdown[i].data, lora=site("down"), t=ops.decode_lora_t(hm, sd.A.data, 16 * sd.n_sub, sd.scale))
        hf, _, _ = ops.rmsnorm_fwd(x, self.norm.data, c.rms_norm_eps, residual=delta, gemma=self.gemma)
        return hf

    @torch.no_grad()
    def _next_token(self, logits, do_sample, temperature, top_k, top_p, generator):
        """GenerationMixin._sample's token choice (generation/utils.py:3131-3250): argmax, or -- do_sample -- a draw from the
        distribution after TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper (generation/logits_process.py), in that order.
        [B, V] fp32 logits -> [B] int64.  (A few torch calls on a [B, V] tensor per token: bookkeeping, not the hot path.)"""
        if not do_sample:
            return logits.argmax(-1)
        scores = logits / temperature if temperature != 1.0 else logits.clone()
        if top_k and 0 < top_k < scores.shape[-1]:
            kth = torch.topk(scores, top_k, dim=-1).values[..., -1, None]
            scores = scores.masked_fill(scores < kth, float("-inf"))
        if top_p is not None and top_p < 1.0:
            srt, idx = torch.sort(scores, descending=False, dim=-1)
            remove = srt.softmax(-1).cumsum(-1) <= (1.0 - top_p)
            remove[..., -1:] = False                                       # min_tokens_to_keep = 1
            scores = scores.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
        return torch.multinomial(scores.softmax(-1), 1, generator=generator).squeeze(-1)

    @torch.no_grad()
    def _merged_weights(self):
        """W + (alpha / r) B A of every adapted projection -- peft's merge_and_unload() -- in buffers that live on the model (the captured decode step of the merged
        mode points at them: they are rewritten in place on every call, since the optimizer updates the adapters through raw pointers and no version counter sees it).
        The update B A is formed by the library's own product (bf16 out, K = 64) and added in fp32, rounded once: NOT bit-identical to the unmerged branch, whose
        t = A x is rounded to bf16 before it meets B."""
        bufs = self.__dict__.get("_merged")
        names = (("qkv", self.wqkv), ("o", self.wo), ("gu", self.wgu), ("down", self.wdown))
        if bufs is None:
            bufs = self.__dict__["_merged"] = {k: [torch.empty_like(w.data) for w in ws] for k, ws in names}
        for k, ws in names:
            for i, w in enumerate(ws):
                site = self.lora[i][k]
                delta = ops.gemm_nt(site.B.data, ops.transpose(site.A.data), alpha=site.scale)      # [out, in]: block b's rows only see its own rank-r rows of A
                bufs[k][i].copy_(w.data.float() + delta.float())
        return bufs

    def generate(self, input_ids=None, attention_mask=None, max_new_tokens=128, pad_token_id=None, eos_token_id=None,
                 use_cache=True, return_logits=False, use_graph=None, do_sample=None, temperature=None, top_k=None, top_p=None,
                 generator=None, merge_adapters=False, **kw):
        """merge_adapters=True (opt-in, LoRA models only): the adapters are folded into copies of the projection weights once per call and the decode step runs
        without the five adapter branches per layer -- peft's `merge_and_unload()` for inference.  Not bit-identical to the unmerged step (see `_merged_weights`);
        the reference's own eval loop (ecg_byte/runners/inference.py, main.py:181-195) generates with the adapters unmerged, which stays the default."""
        if merge_adapters and self.lora is not None:
            merged = self._merged_weights()
            params = {"qkv": self.wqkv, "o": self.wo, "gu": self.wgu, "down": self.wdown}
            saved_lora, saved = self.lora, {k: [w.data for w in ws] for k, ws in params.items()}
            try:
                self.lora = None
                for k, ws in params.items():
                    for w, mw in zip(ws, merged[k]):
                        w.data = mw
                return self._generate(input_ids, attention_mask, max_new_tokens, pad_token_id, eos_token_id, use_cache, return_logits, use_graph, do_sample,
                                      temperature, top_k, top_p, generator)
            finally:
                self.lora = saved_lora
                for k, ws in params.items():
                    for w, d in zip(ws, saved[k]):
                        w.data = d
        return self._generate(input_ids, attention_mask, max_new_tokens, pad_token_id, eos_token_id, use_cache, return_logits, use_graph, do_sample,
                              temperature, top_k, top_p, generator)

    def _generate(self, input_ids=None, attention_mask=None, max_new_tokens=128, pad_token_id=None, eos_token_id=None,
                  use_cache=True, return_logits=False, use_graph=None, do_sample=None, temperature=None, top_k=None, top_p=None,
                  generator=None):
        """GenerationMixin.generate / _sample (generation/utils.py:1877, 3131-3250) as LLM.generate calls it
        (ecg_byte/models/llm.py:26-37): positions from the attention mask (utils.py:410-411), finished sequences keep emitting
        pad_token_id, stop when every sequence has produced eos_token_id or after max_new_tokens.  Greedy unless do_sample: the
        reference's call passes no sampling arguments, so HF falls back to the checkpoint's generation_config.json -- which
        the Llama-3.2 conversion writes with do_sample=True, temperature 0.6, top_p 0.9 (convert_llama_weights_to_hf.py:402-408);
        `from_pretrained` reads that file into `self.generation_config` and the arguments default to it.
        use_graph: None (default) = the decode step runs as a replayed HIP graph (`_generate_graph`) whenever nothing asks for the eager loop
        (per-step logits, no cache, sampling); True / False force it.
        Returns [B, S0 + generated] int64 (prompt included)."""
        gc = getattr(self, "generation_config", None) or {}
        do_sample = bool(gc.get("do_sample", False)) if do_sample is None else bool(do_sample)
        temperature = float(gc.get("temperature", 1.0)) if temperature is None else float(temperature)
        top_k = int(gc.get("top_k", 50)) if top_k is None else int(top_k)                    # GenerationConfig's defaults
        top_p = float(gc.get("top_p", 1.0)) if top_p is None else float(top_p)
        V = self.cfg.vocab_size
        if do_sample:
            pick = lambda lg16: self._next_token(lg16[:, :V].float(), True, temperature, top_k, top_p, generator)
        else:
            pick = lambda lg16: ops.argmax_rows(lg16, V)        # the bf16 logits as the head GEMM leaves them: one launch, first maximum (torch.argmax's rule)
        c = self.cfg
        dev = self.device
        input_ids = input_ids.to(dev).long()
        B, S0 = input_ids.shape
        cap = S0 + max_new_tokens
        mask = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        mask[:, :S0] = attention_mask.to(dev).float() if attention_mask is not None else 1.0
        eos = None
        if eos_token_id is not None:
            eos = torch.as_tensor(eos_token_id, device=dev).view(-1)
            if pad_token_id is None:
                pad_token_id = int(eos[0])                                 # utils.py: "Setting pad_token_id to eos_token_id"

        def positions(m):
            p = m.long().cumsum(-1) - 1
            return p.masked_fill(m == 0, 1)

        if use_graph is None:
            use_graph = generator is None and not do_sample          # (sampling replays too when asked: use_graph=True)
        if use_cache and use_graph and not return_logits and max_new_tokens > 2:
            return self._generate_graph(input_ids, mask, S0, max_new_tokens, pad_token_id, eos, positions, pick,
                                        (do_sample, temperature, top_k, top_p) if do_sample else None)
        width = 2 * c.num_key_value_heads * c.head_dim
        cap_rows = -(-cap // 128) * 128                                   # as _generate_graph sizes them: the same split of the keys, the same bits
        caches = [torch.empty((B, cap_rows, width), dtype=torch.bfloat16, device=dev) for _ in range(c.num_hidden_layers)] if use_cache else None
        if cap_rows != cap:
            mask = torch.cat([mask, mask.new_zeros((B, cap_rows - cap))], 1)
        seq = input_ids
        unfinished = torch.ones(B, dtype=torch.long, device=dev)
        step_logits = []
        for t in range(max_new_tokens):
            n = S0 + t                                                       # tokens in `seq`
            if t == 0 or not use_cache:
                m = mask[:, :n].contiguous()
                hf = self._hidden_states(seq, m, positions(m), caches)
                last = hf.view(B, n, -1)[:, -1].contiguous()
            else:
                pos = positions(mask[:, :n])[:, -1]
                last = self._decode_step(seq[:, -1].contiguous(), pos, mask, caches, n)
            logits = ops.gemm_nt(last, self.embed.data)
            if return_logits:
                step_logits.append(logits[:, :c.vocab_size].float())
            nxt = pick(logits)
            if eos is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
            seq = torch.cat([seq, nxt[:, None]], 1)
            mask[:, n] = 1.0
            if eos is not None:
                unfinished = unfinished * (~torch.isin(nxt, eos)).long()
                if int(unfinished.max()) == 0:
                    break
        return (seq, torch.stack(step_logits, 1)) if return_logits else seq

    def _generate_graph(self, seq, mask, S0, max_new_tokens, pad_token_id, eos, positions, pick, sampling=None):
        """The generate loop with the decode step captured once in a HIP graph and replayed per token.  Everything that changes from step to
        step lives in device memory -- the token, its position, the number of valid cache rows, the column the next token goes to, the
        `unfinished` flags -- so the host only replays and, when an eos id is given, reads one flag per token.
        The captured step is kept on the model and REUSED by later calls (`self._gen_graphs`): its buffers (KV caches, mask, output) are sized
        to the call's prompt + new tokens rounded up to 128 rows, so the eval loop's calls (one per test sample, prompts of different lengths,
        ecg_byte/runners/tester.py) replay one graph; a call writes its prompt state into the buffers and replays.  Key: batch, rounded
        capacity, pad / eos ids, the sampling arguments, and the storage of the embedding table and of the adapters (a resized vocabulary
        or enable_lora() captures anew).  At Gemma-2B dims with adapters (about 400 launches per token, which the host issues at 8 us each):
        3.5 ms per token eagerly, 2.25 replayed (423 tokens/s after a 600-token prompt; 530 without adapters); capturing costs about 16 ms, once.
        The greedy choice inside the step is ecgb_argmax_bf16, not torch.argmax: over a 260 000-wide row torch's reduction is two passes over a
        semaphore buffer, and a graph holding it, replayed after other eager reductions had run, left the token unwritten (measured: the second
        call of a reused graph read 8 bytes of RoPE table as a token id) -- the step must not depend on state outside its own buffers."""
        c = self.cfg
        dev = self.device
        B = seq.shape[0]
        cap = -(-(S0 + max_new_tokens) // 128) * 128
        pad_id = pad_token_id if pad_token_id is not None else 0
        key = (B, cap, pad_id, None if eos is None else tuple(int(e) for e in eos.tolist()), sampling, self.training, self.embed.data_ptr(),
               None if self.lora is None else self.lora[0]["qkv"].A.data_ptr(), self.wqkv[0].data_ptr(), bool(self.decode_attn_one), bool(self.decode_lora_one))      # (the last: base weights or their merged copies)
        graphs = self.__dict__.setdefault("_gen_graphs", {})
        st = graphs.get(key) if sampling is None else None                   # (a sampling step holds torch's sort / cumsum / multinomial: captured per call, see below)
        if st is None:
            while len(graphs) >= 4:                                          # a handful of shapes at most: drop the oldest
                graphs.pop(next(iter(graphs)))
            width = 2 * c.num_key_value_heads * c.head_dim
            st = SimpleNamespace(graph=None,
                                 caches=[torch.empty((B, cap, width), dtype=torch.bfloat16, device=dev) for _ in range(c.num_hidden_layers)],
                                 mask=torch.zeros((B, cap), dtype=torch.float32, device=dev),
                                 out=torch.zeros((B, cap), dtype=torch.long, device=dev),
                                 tok=torch.zeros(B, dtype=torch.long, device=dev), pos=torch.zeros(B, dtype=torch.long, device=dev),
                                 n_dev=torch.zeros(1, dtype=torch.int32, device=dev), col=torch.zeros((B, 1), dtype=torch.long, device=dev),
                                 unfinished=torch.ones(B, dtype=torch.long, device=dev),
                                 ones_col=torch.ones((B, 1), dtype=torch.float32, device=dev),
                                 pad_t=torch.full((B,), pad_id, dtype=torch.long, device=dev), eos=eos, scratch=None)
            ns = ops.decode_splits(cap, B, c.num_attention_heads)
            st.scratch_one = None
            if self.decode_attn_one and ops.decode_one_ok(B, c.num_attention_heads, c.head_dim, ns, cap):
                st.scratch_one = ops.decode_one_scratch(B, c.num_attention_heads, c.head_dim, ns, dev)
            elif ns > 1:
                st.scratch = ops.decode_split_scratch(cap, B, c.num_attention_heads, c.head_dim, ns, dev)
            if sampling is None:
                graphs[key] = st
        caches, gmask, out, tok, pos, n_dev, col, unfinished = st.caches, st.mask, st.out, st.tok, st.pos, st.n_dev, st.col, st.unfinished
        # prefill + first token, eagerly
        gmask.zero_()
        gmask[:, :S0] = mask[:, :S0]
        m0 = gmask[:, :S0].contiguous()
        hf = self._hidden_states(seq, m0, positions(m0), caches)
        logits = ops.gemm_nt(hf.view(B, S0, -1)[:, -1].contiguous(), self.embed.data)
        unfinished.fill_(1)
        nxt = pick(logits)
        if eos is not None:
            unfinished.mul_((~torch.isin(nxt, eos)).long())
        out.fill_(pad_id)
        out[:, :S0] = seq
        out[:, S0] = nxt
        gmask[:, S0] = 1.0
        if (eos is not None and int(unfinished.max()) == 0) or max_new_tokens == 1:
            return out[:, :S0 + 1].clone()
        # state of the captured step
        tok.copy_(nxt)
        pos.copy_(positions(gmask[:, :S0 + 1])[:, -1])                       # position of the token in `tok`
        n_dev.fill_(S0 + 1)                                                  # cache rows valid once `tok` is appended
        col.fill_(S0 + 1)                                                    # where the token produced by the step goes
        eos_s, ones_col, pad_t = st.eos, st.ones_col, st.pad_t

        eos_i64 = None if eos_s is None else eos_s.to(torch.int64).contiguous()

        def step():
            one = self.lora is not None and not self.training and self.decode_lora_one
            last = self._decode_step(tok, pos, gmask, caches, None, n_dev, scratch=st.scratch, scratch_one=st.scratch_one, epoch_advanced=one)
            nx = pick(ops.gemm_nt(last, self.embed.data))
            # pad for finished sequences, the eos test, the token into `out` and `tok`, the mask's new column, the three counters: one launch (six to eleven
            # element-wise ones before, 5 us each in the replayed graph)
            ops.decode_advance_(nx.to(torch.int64), tok, pos, col, n_dev, out, gmask, unfinished, pad_id, eos_i64, epoch=self._decode_epoch(dev) if one else None)

        if st.graph is None:
            state = (tok, pos, n_dev, col, unfinished, out, gmask)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                  # warm-up outside the capture (allocations, lazy init)
                snap = tuple(t.clone() for t in state)
                step()
                for dst, src in zip(state, snap):
                    dst.copy_(src)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            for dst, src in zip(state, snap):                              # capture does not execute, but keep the state explicit
                dst.copy_(src)
            st.graph = graph
        produced = 1
        for t in range(1, max_new_tokens):
            st.graph.replay()
            produced += 1
            if eos is not None and int(unfinished.max()) == 0:
                break
        return out[:, :S0 + produced].clone()

    # ---- backward -------------------------------------------------------------------------------
    def _backward(self, grad_out):
        if self.grad_sync is not None and hasattr(self.grad_sync, "backward_kernels"):
            with self.grad_sync.backward_kernels():     # one-tile input-gradient GEMMs only while a gradient exchange can be in flight
                return self._backward_impl(grad_out)
        return self._backward_impl(grad_out)

    def _backward_impl(self, grad_out):
        self.sync_optimizer()                                        # gradients are about to be overwritten; the adapters' shadows are rebuilt from the updated weights
        c = self.cfg
        H, I, D, Hq, Hkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        G = Hq // Hkv
        QKV = self.qkv
        saved, input_ids, mask, cos, sin, (xf, rstdf), dhf, (B, S) = self._saved
        self._saved = None
        dev = self.device
        T = B * S
        scale = 1.0 / math.sqrt(D)
        go = float(grad_out)   # d(final)/d(loss); 1.0 for loss.backward()
        frozen = self.lora is not None
        if frozen:
            self._refresh_lora_shadows()
        if go != 1.0:
            dhf = (dhf.float() * go).to(torch.bfloat16)
            if not frozen:
                self.embed_grad_head.copy_((self.embed_grad_head.float() * go).to(torch.bfloat16))

        self._grad_layout()
        L = c.num_hidden_layers
        # Data parallel: the tied head's share of the embedding gradient (dense, 0.54 GB at Llama-3.2-1B's vocabulary) was finished by the loss head in
        # the forward pass -- it leaves NOW, under the whole backward; what the input rows add at the very end travels as (ids, rows) (see _embedding_grad_sparse)
        sparse_embed = (not frozen and self.grad_sync is not None and getattr(self.grad_sync, "sparse_embedding", False)
                        and self.embed.grad is None)
        if sparse_embed:
            view, _ = self._grad_slot(self.embed)
            if self.embed_grad_head.data_ptr() != view.data_ptr():
                view.copy_(self.embed_grad_head)
            self.embed.grad = view
            off = (view.data_ptr() - self._gflat.data_ptr()) // 2
            self.grad_sync.on_flat_ready(self._gflat, off, off + view.numel())
            self.grad_sync._flush()

        def wgrad(dy, xin, param):
            """param.grad = dy^T . xin (bf16): contraction over the token rows of both operands, no transposed copies"""
            if not frozen:
                self._wgrad(param, dy, xin)

        def lngrad(param, dw32):
            if not frozen:
                self._vgrad(param, dw32)

        # fp32 rows for the norm weights' gradients: ONE zero fill per step (a fill per norm was 33 launches of a few microseconds each on a queue that is never empty);
        # frozen norm weights (LoRA): no weight gradient is computed
        dw_rows = None if frozen else iter(torch.zeros((2 * c.num_hidden_layers + 1, H), dtype=torch.float32, device=dev).unbind(0))
        dw = None if frozen else next(dw_rows)
        g = ops.rmsnorm_bwd(xf, self.norm.data, rstdf, dhf, dw, gemma=self.gemma)          # grad of the residual stream
        lngrad(self.norm, dw)
        for i in reversed(range(c.num_hidden_layers)):
            x1, rstd1, h1, qkv, P, ao, x2, rstd2, h2, gu, hm, ls = saved.pop()
            # MLP
            wgrad(g, hm, self.wdown[i])
            if not frozen and ops.nn_glu_bwd_eligible(g.shape[0], I, H):
                d_gu = ops.gemm_nn_glu_bwd(g, self.wdown[i].data, gu, gelu_tanh=self.gemma)       # dX of the down projection + GLU backward, one launch
            else:
                if frozen:                                         # adapters + GLU backward in one pass (with the product itself where the kernel takes the shape)
                    d_gu = self.lora[i]["down"].backward(g, ls[3], self, lambda: self._dx(g, ("wdown", i), self.wdown[i]), glu=(gu, self.gemma), weight=self.wdown[i].data)
                else:
                    d_hm = self._dx(g, ("wdown", i), self.wdown[i])    # [T, I]
                    d_gu = ops.glu_bwd(gu, d_hm, gelu_tanh=self.gemma)
                    del d_hm
            del hm
            wgrad(d_gu, h2, self.wgu[i])
            d_h2 = self._dx(d_gu, ("wgu", i), self.wgu[i])         # [T, H]
            if frozen:
                self.lora[i]["gu"].backward(d_gu, ls[2], self, d_h2)
            del d_gu, gu
            dw = None if frozen else next(dw_rows)
            g2 = ops.rmsnorm_bwd(x2, self.ln2[i].data, rstd2, d_h2, dw, dres=g, gemma=self.gemma)
            lngrad(self.ln2[i], dw)
            # attention output projection
            wgrad(g2, ao, self.wo[i])
            if frozen:                                             # [T, Hq*D]: product and adapter share in one launch where the kernel takes the shape
                d_ao = self.lora[i]["o"].backward(g2, ls[1], self, lambda: self._dx(g2, ("wo", i), self.wo[i]), weight=self.wo[i].data)
            else:
                d_ao = self._dx(g2, ("wo", i), self.wo[i])
            # attention core
            if self.fused_attention:
                d_qkv = ops.attn_bwd(qkv, mask, ao, d_ao, P, B, S, Hq, Hkv, D, scale, rope=(cos, sin))   # RoPE's backward inside (head_dim 64) or behind it
            else:
                d_qkv = self._attn_materialised_bwd(qkv, d_ao, P, B, S)
                ops.rope_(d_qkv, cos, sin, Hq + Hkv, D, QKV, inverse=True)
            del P
            wgrad(d_qkv, h1, self.wqkv[i])
            bottom = frozen and i == 0                             # frozen embeddings below: nobody needs the gradient of the first layer's input
            d_h1 = None if bottom else self._dx(d_qkv, ("wqkv", i), self.wqkv[i])      # [T, H]   (autograd would not compute it either)
            if frozen:
                self.lora[i]["qkv"].backward(d_qkv, ls[0], self, d_h1)
            if not bottom:
                dw = None if frozen else next(dw_rows)
                g = ops.rmsnorm_bwd(x1, self.ln1[i].data, rstd1, d_h1, dw, dres=g2, gemma=self.gemma)
                lngrad(self.ln1[i], dw)
            if self.grad_sync is not None:   # this layer's gradients are final: its range of the flat buffer may leave
                self.grad_sync.on_flat_ready(self._gflat, *self._granges[L - 1 - i])
        if not frozen:
            pad = c.pad_token_id if c.pad_token_id is not None else -1       # nn.Embedding(padding_idx): no lookup gradient for that row
            # Rows that are masked AND carry no loss have a gradient of exactly zero (never attended as keys, zero attention output as queries, no loss of their own):
            # skipped as one run.  A masked row whose shifted label is valid (a hole in the mask, the last pad row in front of a labelled first token) still receives
            # d hidden through its residual path, as HF autograd gives it: it stays.
            dead = (mask.reshape(-1) == 0) & self._unlabelled_rows
            live_ids = input_ids.view(-1).masked_fill(dead, -1)
            if sparse_embed:
                self._embedding_grad_sparse(live_ids, g, pad)
            else:
                self._embedding_grad(self.embed, self.embed_grad_head, live_ids, g, self.embed_scale, pad)
        if self.grad_sync is not None:
            if not frozen:
                lo, hi = self._granges[L]
                if sparse_embed:                                  # the table's slice has been exchanged already: only what lies behind it (the final norm)
                    lo = (self._gview[id(self.embed)].data_ptr() - self._gflat.data_ptr()) // 2 + (self.embed.numel() + 127) // 128 * 128
                self.grad_sync.on_flat_ready(self._gflat, lo, hi)
            self.grad_sync.finish()

    # ---- fused optimizer (clip_grad_norm_(1.0) + Adam with L2, Noam schedule) -------------------------
    def make_optimizer(self, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, warmup=500, max_norm=1.0, overlap=False):
        return HipAdam(self, lr, betas, eps, weight_decay, warmup, max_norm, overlap=overlap)


class HipAdam:
    """ecg_byte/runners/train.py:26 + ecg_byte/main.py:262-264 + ecg_byte/scheduler.py:3-27 on the device:
    global-norm clip to 1.0, Adam (weight decay as L2), lr = d_model^-0.5 * min(t^-0.5, t * warmup^-1.5).
    Moments are fp32 (the reference's follow the bf16 parameter dtype)."""

    def __init__(self, model: HipCausalLM, lr, betas, eps, weight_decay, warmup, max_norm, overlap=False):
        self.model, self.betas, self.eps, self.wd, self.warmup, self.max_norm = model, betas, eps, weight_decay, warmup, max_norm
        # overlap=True: the parameter updates run on a side HIP stream, group by group in the order the forward pass reads them, and the NEXT forward
        # waits per group (an event each): the step's 27 GB of moment / weight traffic (5.7 ms at the HBM rate) hides under the next forward's GEMMs,
        # which leave the memory system idle.  The gradient norm (needs every gradient) stays in front.  Same arithmetic, same bits.  Anything that
        # touches parameters through the model (backward, inference, state_dict, generate) waits for the whole step first (`sync_optimizer`); code
        # that reads `p.data` directly right after `step()` must call `model.sync_optimizer()` itself -- hence opt-in (bench.py, main.py).
        self.overlap = bool(overlap) and getattr(model, "supports_optimizer_overlap", False) and hasattr(model, "_opt_groups") and torch.cuda.is_available()
        self._side = None
        self._acc = None
        self.init_lr = model.cfg.hidden_size ** -0.5
        self.fixed_lr = None if warmup else lr
        self.t = 0
        self.state = {}
        self._sumsq_plan = None
        self._adam_tables = None

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None

    def lr(self, t):
        if self.fixed_lr is not None:
            return self.fixed_lr
        return self.init_lr * min(t ** -0.5, t * self.warmup ** -1.5)

    def step_and_update_lr(self):
        self.t += 1
        params = [p for p in self.model.parameters() if p.grad is not None]
        acc = torch.zeros(1, dtype=torch.float32, device=self.model.device)
        grads = [p.grad for p in params]
        if all(g.dtype == torch.bfloat16 and g.is_contiguous() for g in grads):
            counts = [g.numel() for g in grads]
            if self._sumsq_plan is None or self._sumsq_plan.counts_host != counts:
                self._sumsq_plan = ops.SumsqPlan(counts, self.model.device)
            ops.sumsq_multi(grads, acc, self._sumsq_plan)       # one launch for all gradients
        else:
            for g in grads:
                ops.sumsq(g, acc)
        lr = self.lr(self.t)

        def update(p):
            st = self.state.get(id(p))
            if st is None or st[0].shape != p.shape:
                st = (torch.zeros(p.shape, dtype=torch.float32, device=p.device), torch.zeros(p.shape, dtype=torch.float32, device=p.device))
                self.state[id(p)] = st
            ops.adam_step_(p.data, p.grad, st[0], st[1], acc, self.max_norm, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t)

        if not self.overlap:
            multi = self._sumsq_plan is not None and self._sumsq_plan.counts_host == [p.numel() for p in params] and \
                all(p.data.dtype == torch.bfloat16 and p.data.is_contiguous() and p.grad.dtype == torch.bfloat16 and p.grad.is_contiguous() for p in params)
            if multi:                                                # ONE launch for every parameter (was one per tensor: ~100 for a full fine-tune)
                for p in params:
                    st = self.state.get(id(p))
                    if st is None or st[0].shape != p.shape:
                        self.state[id(p)] = (torch.zeros(p.shape, dtype=torch.float32, device=p.device), torch.zeros(p.shape, dtype=torch.float32, device=p.device))
                datas, ms, vs = [p.data for p in params], [self.state[id(p)][0] for p in params], [self.state[id(p)][1] for p in params]
                if self._adam_tables is None or not self._adam_tables.matches(datas, grads, ms, vs):
                    self._adam_tables = ops.AdamMultiPlan(datas, grads, ms, vs, self.model.device)
                ops.adam_multi_(self._adam_tables, self._sumsq_plan, acc, self.max_norm, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t)
            else:
                for p in params:
                    update(p)
        else:
            model = self.model
            model.sync_optimizer()                                   # (a previous overlapped step nobody waited for)
            if self._side is None:
                self._side = torch.cuda.Stream(device=model.device)
            main = torch.cuda.current_stream()
            for p in params:                                         # moments are created on the main stream, before the fork
                if id(p) not in self.state or self.state[id(p)][0].shape != p.shape:
                    self.state[id(p)] = (torch.zeros(p.shape, dtype=torch.float32, device=p.device), torch.zeros(p.shape, dtype=torch.float32, device=p.device))
            acc.record_stream(self._side)
            fork = torch.cuda.Event()
            fork.record(main)
            self._side.wait_event(fork)
            have = {id(p) for p in params}
            with torch.cuda.stream(self._side):
                seen = set()
                for key, group in model._opt_groups():
                    for p in group:
                        if id(p) in have:
                            update(p)
                            seen.add(id(p))
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    model._opt_ready[key] = ev
                for p in params:                                     # anything the groups do not name
                    if id(p) not in seen:
                        update(p)
                done = torch.cuda.Event()
                done.record(self._side)
                model._opt_done = done
        self.model._drop_shadows({id(p) for p in params})   # updated weights: their transposed shadows are stale

    step = step_and_update_lr
