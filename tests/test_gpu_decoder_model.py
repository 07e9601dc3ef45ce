"""GPU test of the whole decoder step (HipCausalLM) against goldens generated from the VENDORED
transformers LlamaForCausalLM (tests/golden/make_decoder_golden.py): tiny config, GQA, llama3 RoPE
scaling, left-padded batch, -100 labels.  Tolerances (bf16 compute vs the fp32 reference run):
loss within 1e-2 relative (the reference's own bf16 run differs from its fp32 run by ~1e-4 here);
every parameter gradient within 3e-2 of the fp32 gradient in relative Frobenius norm and with
cosine similarity > 0.999."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _load():
    z = np.load(os.path.join(GOLDEN, "decoder_llama_tiny.npz"))
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    cfg = DecoderConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=500000.0, pad_token_id=299,
                        rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                      "original_max_position_embeddings": 32, "rope_type": "llama3"})
    m = HipCausalLM(cfg)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m.load_state_dict(sd)
    return z, m


def _batch(z):
    return dict(input_ids=torch.from_numpy(z["input_ids"]).cuda(), attention_mask=torch.from_numpy(z["attention_mask"]).cuda(),
                labels=torch.from_numpy(z["labels"]).cuda(), position_ids=torch.from_numpy(z["position_ids"]).cuda())


def test_rope_frequencies_match_reference():
    z, m = _load()
    assert np.allclose(m.inv_freq.cpu().numpy(), z["inv_freq"], rtol=1e-6, atol=0)


def test_state_dict_roundtrip_hf_names():
    z, m = _load()
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("w:"):
            assert torch.equal(sd[k[2:]].float().cpu(), torch.from_numpy(z[k])), k
    assert "lm_head.weight" in sd and torch.equal(sd["lm_head.weight"], sd["model.embed_tokens.weight"])


def _compare_grads(z, m):
    c = m.cfg
    D, Hq, Hkv, I = c.head_dim, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size
    grads = {"model.embed_tokens.weight": m.embed.grad[: c.vocab_size], "model.norm.weight": m.norm.grad}
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        g = m.wqkv[i].grad
        grads[p + "self_attn.q_proj.weight"] = g[: Hq * D]
        grads[p + "self_attn.k_proj.weight"] = g[Hq * D: Hq * D + Hkv * D]
        grads[p + "self_attn.v_proj.weight"] = g[Hq * D + Hkv * D:]
        grads[p + "self_attn.o_proj.weight"] = m.wo[i].grad
        grads[p + "mlp.gate_proj.weight"] = m.wgu[i].grad[:I]
        grads[p + "mlp.up_proj.weight"] = m.wgu[i].grad[I:]
        grads[p + "mlp.down_proj.weight"] = m.wdown[i].grad
        grads[p + "input_layernorm.weight"] = m.ln1[i].grad
        grads[p + "post_attention_layernorm.weight"] = m.ln2[i].grad
    for name, g in grads.items():
        ref = torch.from_numpy(z["g:" + name]).cuda()
        g = g.float()
        rel = (g - ref).norm() / ref.norm().clamp_min(1e-12)
        cos = torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0)
        assert rel.item() < 3e-2 and cos.item() > 0.999, (name, rel.item(), cos.item())


@pytest.mark.parametrize("fused_attention", [True, False], ids=["fused-attn", "materialised-scores"])
@pytest.mark.parametrize("full_logits", [False, True])
def test_loss_and_gradients_vs_vendored_transformers(full_logits, fused_attention):
    z, m = _load()
    m.full_logits = full_logits
    m.fused_attention = fused_attention
    out = m(**_batch(z))
    loss = out.loss
    assert abs(loss.item() - float(z["loss_fp32"])) <= 1e-2 * float(z["loss_fp32"]), (loss.item(), float(z["loss_fp32"]))
    loss.backward()
    _compare_grads(z, m)


def test_training_steps_reduce_loss_and_follow_torch_adam():
    """Three optimizer steps with the fused HipAdam on the tiny model: loss goes down, and the first
    step equals clip_grad_norm_(1.0) + torch.optim.Adam(L2) applied to the same gradients."""
    z, m = _load()
    opt = m.make_optimizer(warmup=500)
    batch = _batch(z)
    losses = []
    for step in range(3):
        opt.zero_grad()
        out = m(**batch)
        out.loss.backward()
        if step == 0:
            p0 = m.wo[0].data.float().clone()
            g0 = [p.grad.float().clone() for p in m.parameters() if p.grad is not None]
            ref = p0.clone().requires_grad_(True)
            total = torch.sqrt(sum((g * g).sum() for g in g0))
            ref.grad = m.wo[0].grad.float() * min(1.0, 1.0 / (total.item() + 1e-6))
            topt = torch.optim.Adam([ref], lr=opt.lr(1), betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
            topt.step()
        opt.step_and_update_lr()
        if step == 0:
            assert torch.allclose(m.wo[0].data.float(), ref.detach().to(torch.bfloat16).float(), atol=2e-3, rtol=2e-2)
        losses.append(out.loss.item())
    assert losses[2] < losses[0], losses


def test_resize_token_embeddings_and_pad_rows():
    z, m = _load()
    old = m.embed.data[:300].clone()
    m.resize_token_embeddings(300 + 256 + 50 + 3)          # main.py:144-151: signal tokens + 3 specials
    assert m.cfg.vocab_size == 609 and m.embed.shape[0] % 128 == 0
    assert torch.equal(m.embed.data[:300], old)
    assert torch.allclose(m.embed.data[300:609].float(), old.float().mean(0).to(torch.bfloat16).float()[None].expand(309, -1))
    b = _batch(z)
    b["input_ids"] = b["input_ids"].clone()
    b["input_ids"][1, 5] = 600
    out = m(**b)
    out.loss.backward()
    assert torch.isfinite(out.loss) and m.embed.grad[600].abs().sum() > 0


@pytest.mark.parametrize("fused_attention", [True, False], ids=["fused-attn", "materialised-scores"])
def test_gqa_4_to_1_vs_decoder_oracle(fused_attention):
    """A second shape (4 query heads per KV head, 3 layers, S = 128, default RoPE) against the
    PyTorch oracle run in fp32 on the same device; exercises the per-group dK/dV accumulation."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    from oracle import llama_ref as R
    cfgd = dict(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=3, num_attention_heads=4,
                num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-6)
    params = R.random_params(cfgd, seed=3, device="cuda", std=0.05)
    cfg = DecoderConfig(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=3, num_attention_heads=4,
                        num_key_value_heads=1, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=514)
    m = HipCausalLM(cfg)
    m.fused_attention = fused_attention
    m.load_state_dict(params)
    B, S = 2, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    ids = torch.randint(0, 514, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda"); mask[1, :70] = 0; ids[1, :70] = 514
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device="cuda"); labels[:, -30:] = ids[:, -30:]
    ref_p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = R.llama_loss(ref_p, cfgd, ids, mask, labels, pos, R.llama3_inv_freq(64, 10000.0, None).cuda())
    ref.backward()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item()
    Hq, Hkv, D = 4, 1, 64
    for i in range(3):
        got = m.wqkv[i].grad.float()
        want = torch.cat([ref_p[f"model.layers.{i}.self_attn.{n}_proj.weight"].grad for n in "qkv"], 0)
        rel = (got - want).norm() / want.norm()
        assert rel.item() < 3e-2, (i, rel.item())
        for name, gp in (("mlp.down_proj.weight", m.wdown[i].grad), ("self_attn.o_proj.weight", m.wo[i].grad)):
            want = ref_p[f"model.layers.{i}.{name}"].grad
            assert ((gp.float() - want).norm() / want.norm()).item() < 3e-2, (i, name)
    want = ref_p["model.embed_tokens.weight"].grad
    assert ((m.embed.grad[:515].float() - want).norm() / want.norm()).item() < 3e-2


def test_masked_rows_with_a_valid_label_keep_their_embedding_gradient():
    """Backward skips the embedding scatter of rows that are masked AND unlabelled (their gradient is exactly zero).  A masked row whose shifted label is valid --
    a hole in the attention mask in front of a labelled token -- still receives d hidden through its residual path and autograd adds it to the row of its id:
    checked against the fp32 oracle on an id that occurs nowhere else in the batch.  (A left-pad row in front of a labelled FIRST token is the same case for the
    scatter, but its attention row has no visible key at all and the reference's softmax over an all-masked row is not a definition worth matching: not tested.)"""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    from oracle import llama_ref as R
    cfgd = dict(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6)
    params = R.random_params(cfgd, seed=11, device="cuda", std=0.05)
    cfg = DecoderConfig(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=None)
    m = HipCausalLM(cfg)
    m.load_state_dict(params)
    B, S = 2, 128
    g = torch.Generator(device="cuda").manual_seed(6)
    ids = torch.randint(0, 500, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda")
    mask[0, 100] = 0; ids[0, 100] = 510                    # a hole: row 100 is masked, labels[101] is valid
    mask[1, :40] = 0; ids[1, :40] = 511                    # left padding, unlabelled: dead rows
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
    labels = torch.full((B, S), -100, device="cuda")
    labels[0, 90:] = ids[0, 90:]; labels[1, 60:] = ids[1, 60:]
    labels[0, 100] = -100                                  # (the hole itself is no target)
    ref_p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = R.llama_loss(ref_p, cfgd, ids, mask, labels, pos, R.llama3_inv_freq(64, 10000.0, None).cuda())
    ref.backward()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item()
    want, got = ref_p["model.embed_tokens.weight"].grad, m.embed.grad[:515].float()
    # (the head is tied: every vocabulary row also carries the loss head's dE; what the INPUT rows add comes on top of it)
    for tok in (510, 511):                                 # the masked-but-labelled row's id; the dead rows' id (head gradient only)
        assert ((got[tok] - want[tok]).norm() / want[tok].norm()).item() < 5e-2, tok
    m2 = HipCausalLM(cfg); m2.load_state_dict(params)      # the same batch with the hole's id moved elsewhere: row 510 must lose exactly the input row's share
    ids2 = ids.clone(); ids2[0, 100] = 509
    m2(input_ids=ids2, attention_mask=mask, labels=labels, position_ids=pos).loss.backward()
    share = got[510] - m2.embed.grad[510].float()
    assert share.norm().item() > 0.05 * got[510].norm().item()
    assert ((got - want).norm() / want.norm()).item() < 3e-2


def test_lora_adapters_vs_oracle():
    """LoRA mode (the reference's launch mode, main.py:131-155): frozen base, r = 16, alpha = 32, adapters on
    q,k,v,o,gate,up,down; dropout 0 for the parity check.  B is given non-zero values so that every adapter
    gradient is exercised.  peft is not in the tree: the oracle restates y = Wx + (alpha/r) B(Ax)."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    from oracle import llama_ref as R
    cfgd = dict(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6)
    params = R.random_params(cfgd, seed=4, device="cuda", std=0.05)
    cfg = DecoderConfig(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=514)
    m = HipCausalLM(cfg)
    m.load_state_dict(params)
    m.enable_lora(r=16, alpha=32, dropout=0.0, seed=1)
    g = torch.Generator(device="cuda").manual_seed(9)
    ref_p = {k: v.clone() for k, v in params.items()}
    for name, t in m.lora_named():
        if "lora_B" in name:
            t.copy_((torch.randn(t.shape, device="cuda", generator=g) * 0.05).to(torch.bfloat16))
        key = name.replace("base_model.model.", "").replace(".default.weight", "")
        ref_p[key] = t.float().clone().requires_grad_(True)
    B, S = 2, 128
    ids = torch.randint(0, 514, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda"); mask[0, :50] = 0; ids[0, :50] = 514
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device="cuda"); labels[:, -25:] = ids[:, -25:]
    ref = R.llama_loss(ref_p, cfgd, ids, mask, labels, pos, R.llama3_inv_freq(64, 10000.0, None).cuda(), lora_scale=2.0)
    ref.backward()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item()
    assert all(p.grad is None for n, p in m.named_parameters() if "lora" not in n), "base must stay frozen"
    grads = {}
    names = {"qkv": ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], "o": ["self_attn.o_proj"],
             "gu": ["mlp.gate_proj", "mlp.up_proj"], "down": ["mlp.down_proj"]}
    for i, layer in enumerate(m.lora):
        for key, mods in names.items():
            site = layer[key]
            for b, mod in enumerate(mods):
                lo, hi = site.a_rows(b)
                off, w = site.blocks[b]
                grads[f"model.layers.{i}.{mod}.lora_A"] = site.A.grad[lo:hi]
                grads[f"model.layers.{i}.{mod}.lora_B"] = site.B.grad[off: off + w, lo:hi]
            assert float(site.A.grad[16 * site.n_sub:].abs().max()) == 0.0              # the padding of the stacked layout stays inert
            assert float((site.B.grad.float() * (1 - site.bmask.float())).abs().max()) == 0.0   # a block's rows only carry its own columns
    for k, gq in grads.items():
        want = ref_p[k].grad
        rel = (gq.float() - want).norm() / want.norm().clamp_min(1e-12)
        assert rel.item() < 4e-2, (k, rel.item())
    # an optimizer step only moves the adapters; dropout path runs
    opt = m.make_optimizer()
    before = m.wo[0].data.clone()
    opt.step_and_update_lr()
    assert torch.equal(m.wo[0].data, before)
    m.lora[0]["o"].p = 0.05
    m.train()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert torch.isfinite(out.loss)


def test_lora_state_dict_has_pefts_documented_key_set_and_shapes():
    """Checkpoint interchange, structurally: with `get_peft_model(llm, LoraConfig(r=16, lora_alpha=32, target_modules=[q, v, k, o, gate, down, up]_proj, bias="none"))`
    (ecg_byte/main.py:131-155) peft 0.13's PeftModelForCausalLM.state_dict() holds, as its documentation lays out, every base parameter under `base_model.model.`,
    the wrapped projections' own weights one level down under `.base_layer.`, and per wrapped Linear two adapter matrices `lora_A.default.weight` [r, in_features] and
    `lora_B.default.weight` [out_features, r] (adapter name "default"; no bias entries; embeddings, norms and the tied head are not wrapped).  peft itself is not in
    the tree (parity unpinned: its forward semantics are restated, oracle/llama_ref.py); this pins the names and shapes a reference checkpoint would be read by."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    L, H, I, Hq, Hkv, D, V, r = 2, 256, 448, 4, 2, 64, 515, 16
    cfg = DecoderConfig(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv,
                        rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=V - 1)
    m = HipCausalLM(cfg)
    m.enable_lora(r=r, alpha=32, dropout=0.05)
    want = {"base_model.model.model.embed_tokens.weight": (V, H), "base_model.model.model.norm.weight": (H,), "base_model.model.lm_head.weight": (V, H)}
    proj = {"self_attn.q_proj": (Hq * D, H), "self_attn.k_proj": (Hkv * D, H), "self_attn.v_proj": (Hkv * D, H), "self_attn.o_proj": (H, Hq * D),
            "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}
    for i in range(L):
        pre = f"base_model.model.model.layers.{i}."
        want[pre + "input_layernorm.weight"] = (H,)
        want[pre + "post_attention_layernorm.weight"] = (H,)
        for name, (out_f, in_f) in proj.items():
            want[pre + name + ".base_layer.weight"] = (out_f, in_f)
            want[pre + name + ".lora_A.default.weight"] = (r, in_f)
            want[pre + name + ".lora_B.default.weight"] = (out_f, r)
    sd = m.state_dict()
    assert set(sd) == set(want), (sorted(set(sd) ^ set(want))[:8])
    for k, shape in want.items():
        assert tuple(sd[k].shape) == shape, (k, tuple(sd[k].shape), shape)
    assert all(v.dtype == torch.bfloat16 for v in sd.values())
    assert not any("bias" in k for k in sd)                               # bias="none"
    # peft's initialisation: lora_B zero (the adapted model starts as the base model), lora_A not
    assert all(float(sd[k].float().abs().max()) == 0.0 for k in sd if "lora_B" in k) and all(float(sd[k].float().abs().max()) > 0.0 for k in sd if "lora_A" in k)
    # and back: a peft-spelt checkpoint loads into a fresh adapted model, adapters included
    m2 = HipCausalLM(cfg); m2.enable_lora(r=r, alpha=32, dropout=0.05)
    sd2 = {k: (v + 0.125 if "lora_B" in k else v) for k, v in sd.items()}
    m2.load_state_dict(sd2)
    back = m2.state_dict()
    assert all(torch.equal(back[k], sd2[k]) for k in sd2)


def test_lora_dropout_masks_independent_per_module_and_replayed_in_backward():
    """peft draws one dropout mask PER MODULE (q, k and v are three LoraLayers over the same input): the stacked site's three blocks
    must see three different masks of x at the configured rate, and the backward must apply exactly the masks the forward drew
    (checked against a torch restatement fed the masks the forward kernel stored)."""
    from ecg_byte_amd import decoder_ops as ops
    from ecg_byte_amd.decoder import LoraSite
    T, K, p = 1024, 512, 0.25
    g = torch.Generator().manual_seed(1)
    site = LoraSite(K, [(0, 256), (256, 128), (384, 128)], r=16, alpha=32, dropout=p, device="cuda", gen=g)
    x = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(512, K, device="cuda") * 0.05).to(torch.bfloat16)
    with torch.no_grad():
        site.B.copy_((torch.randn(512, 64, device="cuda") * 0.1).to(torch.bfloat16) * site.bmask)
    y, saved = site.project(x, w, training=True, keep=True)
    _, xd, t, p_used, seed = saved
    keep = (xd != 0) | (x[None] == 0)
    rate = 1.0 - keep.float().mean(dim=(1, 2))
    assert all(abs(float(r) - p) < 0.01 for r in rate), rate                 # every block drops ~p of the elements
    agree = (keep[0] == keep[1]).float().mean().item()                     # independent masks agree on (1-p)^2 + p^2 of them
    assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 0.01, agree
    assert torch.equal(xd, torch.where(keep, x[None].expand_as(xd), torch.zeros_like(xd)))   # masked copies are x or 0, unscaled
    inv = 1.0 / (1.0 - int(p * 65536) / 65536)
    A, B = site.A.data.float(), site.B.data.float()
    t_ref = torch.cat([(xd[f].float() @ A[16 * f: 16 * f + 16].T) for f in range(3)], 1) * (2.0 * inv)
    assert ((t[:, :48].float() - t_ref).norm() / t_ref.norm()).item() < 1e-2 and float(t[:, 48:].abs().max()) == 0.0
    y_ref = x.float() @ w.float().T + t[:, :48].float() @ B[:, :48].T
    assert ((y.float() - y_ref).norm() / y_ref.norm()).item() < 1e-2
    # backward through a stand-in for the model's gradient plumbing
    class _M:
        _t = {}
        def _shadow(self, key, prm): return ops.transpose(prm.data)
        def _wgrad(self, prm, dy, xin, alpha=1.0): prm.grad = ops.gemm_tn(dy, xin, alpha=alpha)
        def _wgrad_rows(self, prm, parts, alpha=1.0):
            gr = torch.zeros_like(prm.data)
            for lo, hi, dy, xin in parts: gr[lo:hi] = ops.gemm_tn(dy, xin, alpha=alpha)
            prm.grad = gr
        def _lora_agrad(self, prm, x, dt, n_sub, n_fields, scale, p, seed):
            gr = torch.zeros_like(prm.data)
            ops.lora_da(x, dt, gr[: 16 * n_sub], n_sub, n_fields, scale, p, seed)
            prm.grad = gr
    dy = torch.randn(T, 512, device="cuda").to(torch.bfloat16)
    dx0 = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    dx = site.backward(dy, saved, _M(), dx0.clone())
    dt = dy.float() @ B                                                   # [T, 64]
    dB_ref = (dy.float().T @ t.float()) * site.bmask.float()
    assert ((site.B.grad.float() - dB_ref).norm() / dB_ref.norm()).item() < 2e-2
    dA_ref = torch.cat([dt[:, 16 * f: 16 * f + 16].T @ xd[f].float() for f in range(3)], 0) * (2.0 * inv)
    assert ((site.A.grad[:48].float() - dA_ref).norm() / dA_ref.norm()).item() < 2e-2
    dx_ref = dx0.float() + sum(keep[f].float() * (dt[:, 16 * f: 16 * f + 16] @ A[16 * f: 16 * f + 16]) for f in range(3)) * (2.0 * inv)
    assert ((dx.float() - dx_ref).norm() / dx_ref.norm()).item() < 1e-2
    y2, saved2 = site.project(x, w, training=True, keep=True)              # a new call draws new masks
    assert not torch.equal(saved2[1], xd)


def _load_generate():
    z, m = _load()
    zg = np.load(os.path.join(GOLDEN, "decoder_generate_tiny.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m.load_state_dict({k: (v * 4.0 if "proj" in k else v) for k, v in sd.items()})   # as make_generate_golden.py does
    m.eval()
    return zg, m


def _check_greedy(seq, logits, want_seq, want_scores, S0, tol):
    """Greedy paths are compared step by step: while the paths agree the bf16 logits must be within `tol` of the
    reference's fp32 scores; a sequence may leave the reference path only at a step where the reference's own top-2
    margin is below `tol` (a tie at bf16 resolution), after which it is no longer comparable.  `tol` is 1.5x the largest
    deviation of the reference's OWN bf16 run from its fp32 scores on this fixture (stored by make_generate_golden.py)."""
    B = seq.shape[0]
    exact = 0
    for b in range(B):
        for t in range(want_scores.shape[1]):
            if S0 + t >= seq.shape[1]:
                break
            ref = want_scores[b, t]
            if np.isfinite(ref).all() and want_seq[b, S0 + t] == ref.argmax():     # an unfinished reference row
                err = np.abs(logits[b, t] - ref).max()
                assert err < tol, (b, t, err)
            if seq[b, S0 + t] != want_seq[b, S0 + t]:
                top2 = np.sort(ref)[-2:]
                assert top2[1] - top2[0] < tol, (b, t, seq[b, S0 + t], want_seq[b, S0 + t], top2)
                break
            exact += 1
    return exact


@pytest.mark.parametrize("use_cache", [True, False], ids=["kv-cache", "recompute"])
def test_generate_greedy_vs_vendored_transformers(use_cache):
    """HipCausalLM.generate against GenerationMixin.generate of the vendored transformers (fixture made by
    tests/golden/make_generate_golden.py): left-padded prompts of length 45, eos reached by one sequence at step 5
    (it must keep emitting pad), 24 new tokens."""
    zg, m = _load_generate()
    ids = torch.from_numpy(zg["input_ids"]).cuda()
    mask = torch.from_numpy(zg["attention_mask"]).cuda()
    S0 = ids.shape[1]
    seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=int(zg["max_new_tokens"]),
                             pad_token_id=int(zg["pad_token_id"]), eos_token_id=int(zg["eos_token_id"]), use_cache=use_cache,
                             return_logits=True)
    seq = seq.cpu().numpy()
    want = zg["sequences"]
    assert (seq[:, :S0] == zg["input_ids"]).all()
    exact = _check_greedy(seq, logits.cpu().numpy(), want, zg["scores"], S0, 1.5 * float(zg["ref_bf16_deviation"].max()))
    assert exact >= 30, exact                                   # 24 + 24 + 6 comparable steps at most
    # the finished sequence pads from the step after its eos
    row = seq[1, S0:]
    k = list(row).index(int(zg["eos_token_id"]))
    assert (row[k + 1:] == int(zg["pad_token_id"])).all() and (want[1, S0:] == row).all()


def test_generate_without_eos_and_logits_forward():
    zg, m = _load_generate()
    ids = torch.from_numpy(zg["input_ids"]).cuda()
    mask = torch.from_numpy(zg["attention_mask"]).cuda()
    S0 = ids.shape[1]
    seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=24, pad_token_id=299, return_logits=True)
    assert seq.shape == (3, S0 + 24)
    exact = _check_greedy(seq.cpu().numpy(), logits.cpu().numpy(), zg["sequences_no_eos"], zg["scores_no_eos"], S0,
                          1.5 * float(zg["ref_bf16_deviation"].max()))
    assert exact >= 40, exact
    # logits-only forward (labels=None) over the prompt: last row equals the first generate step
    pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 1)
    out = m(input_ids=ids, attention_mask=mask, position_ids=pos)
    assert out.loss is None and out.logits.shape == (3, S0, 300)
    assert (out.logits[:, -1] - logits[:, 0]).abs().max().item() < 1e-6


def test_generate_stops_when_every_sequence_finished():
    zg, m = _load_generate()
    ids = torch.from_numpy(zg["input_ids"]).cuda()[1:2]
    mask = torch.from_numpy(zg["attention_mask"]).cuda()[1:2]
    seq = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=24, pad_token_id=299, eos_token_id=int(zg["eos_token_id"]))
    assert seq.shape[1] == ids.shape[1] + 6 and int(seq[0, -1]) == int(zg["eos_token_id"])   # HF returns as soon as all are done


def test_generate_with_merged_adapters_follows_the_unmerged_greedy_path():
    """generate(merge_adapters=True) -- peft's merge_and_unload() for inference: W + (alpha / r) B A folded once per call, the decode step without adapter branches.
    Not bit-identical (the unmerged branch rounds t = A x to bf16 before B): compared step by step with the unmerged run as the greedy fixtures are compared with the
    reference -- logits within bf16 tolerance while the paths agree, a path may only leave at a near-tie.  The replayed graph of the merged mode equals its eager loop,
    and the model's own weights and adapters are untouched afterwards."""
    zg, m = _load_generate()
    m.enable_lora(r=16, alpha=32, dropout=0.05, seed=3)
    g = torch.Generator(device="cuda").manual_seed(4)
    with torch.no_grad():
        for layer in m.lora:
            for site in layer.values():
                site.B.copy_((torch.randn(site.B.shape, device="cuda", generator=g) * 0.05).to(torch.bfloat16) * site.bmask)
    m.eval()
    ids, mask = torch.from_numpy(zg["input_ids"]).cuda(), torch.from_numpy(zg["attention_mask"]).cuda()
    S0 = ids.shape[1]
    kw = dict(input_ids=ids, attention_mask=mask, max_new_tokens=24, pad_token_id=299)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    seq_u, log_u = m.generate(return_logits=True, **kw)
    seq_m, log_m = m.generate(return_logits=True, merge_adapters=True, **kw)
    assert not torch.equal(log_u, log_m)                                   # (a different arithmetic, not the same kernels)
    tol = 3.0 * float(zg["ref_bf16_deviation"].max())
    exact = _check_greedy(seq_m.cpu().numpy(), log_m.cpu().numpy(), seq_u.cpu().numpy(), log_u.cpu().numpy(), S0, tol)
    assert exact >= 40, exact
    # the adapters matter: without them the logits are somewhere else entirely
    lora, m.lora = m.lora, None
    log_0 = m.generate(return_logits=True, **kw)[1]
    m.lora = lora
    assert (log_0[:, 0] - log_u[:, 0]).abs().max().item() > 3 * tol
    # graph replay of the merged mode = its eager loop; the captured merged step is reused and re-merged per call
    a = m.generate(merge_adapters=True, use_graph=True, **kw)
    b = m.generate(merge_adapters=True, use_graph=False, **kw)
    assert torch.equal(a, b) and torch.equal(a, seq_m)
    with torch.no_grad():
        m.lora[0]["o"].B.mul_(-1.0)
    a2 = m.generate(merge_adapters=True, use_graph=True, **kw)
    assert torch.equal(a2, m.generate(merge_adapters=True, use_graph=False, **kw))
    with torch.no_grad():
        m.lora[0]["o"].B.mul_(-1.0)
    after = m.state_dict()
    assert before.keys() == after.keys() and all(torch.equal(before[k], after[k]) for k in before)
    assert torch.equal(m.generate(**kw), seq_u)                            # and the unmerged mode is what it was


# ---- Gemma block (config C5's family): goldens from the vendored GemmaForCausalLM, tests/golden/make_decoder_golden_gemma.py
def _load_gemma():
    z = np.load(os.path.join(GOLDEN, "decoder_gemma_tiny.npz"))
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    cfg = DecoderConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=299,
                        model_type="gemma")
    m = HipCausalLM(cfg)
    assert m.fused_attention and m.gemma and m.embed_scale == float(torch.tensor(128 ** 0.5, dtype=torch.bfloat16))
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    return z, m


@pytest.mark.parametrize("fused_attention", [True, False], ids=["fused-attn", "materialised-scores"])
@pytest.mark.parametrize("full_logits", [False, True])
def test_gemma_loss_and_gradients_vs_vendored_transformers(full_logits, fused_attention):
    """(1 + w) RMSNorm in fp32, gelu-tanh gate, sqrt(hidden) embedding scale, MQA with head_dim 128 != hidden / heads, through the
    fused head_dim-128 attention kernels and through materialised scores: loss within 1e-2 relative of the fp32 reference run,
    every gradient within 3e-2."""
    z, m = _load_gemma()
    m.full_logits = full_logits
    m.fused_attention = fused_attention
    out = m(**_batch(z))
    assert abs(out.loss.item() - float(z["loss_fp32"])) <= 1e-2 * float(z["loss_fp32"]), (out.loss.item(), float(z["loss_fp32"]))
    out.loss.backward()
    _compare_grads(z, m)


@pytest.mark.parametrize("use_cache", [True, False], ids=["kv-cache", "recompute"])
def test_gemma_generate_greedy_vs_vendored_transformers(use_cache):
    """Greedy generate of the Gemma block: prompt length 45 (not a multiple of the 64-row tile) through the fused head_dim-128
    attention, head_dim 128 decode kernel."""
    z, m = _load_gemma()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m.load_state_dict({k: (v * 4.0 if "proj" in k else v) for k, v in sd.items()})   # as make_decoder_golden_gemma.py does
    m.eval()
    ids = torch.from_numpy(z["gen_input_ids"]).cuda()
    mask = torch.from_numpy(z["gen_attention_mask"]).cuda()
    S0 = ids.shape[1]
    seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=16, pad_token_id=299, use_cache=use_cache, return_logits=True)
    exact = _check_greedy(seq.cpu().numpy(), logits.cpu().numpy(), z["gen_sequences"], z["gen_scores"], S0,
                          1.5 * float(z["gen_ref_bf16_deviation"].max()))
    assert exact >= 24, exact


def test_gemma_pretrained_directory(tmp_path):
    z, m = _load_gemma()
    from ecg_byte_amd.decoder import HipCausalLM
    m.save_pretrained(str(tmp_path / "g"))
    cfg = json.load(open(tmp_path / "g" / "config.json"))
    assert cfg["model_type"] == "gemma" and cfg["head_dim"] == 128 and cfg["hidden_activation"] == "gelu_pytorch_tanh"
    m2 = HipCausalLM.from_pretrained(str(tmp_path / "g"))
    assert m2.gemma and m2.fused_attention
    a, b = m.state_dict(), m2.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    out1, out2 = m(**_batch(z)).loss.item(), m2(**_batch(z)).loss.item()
    assert abs(out1 - out2) < 1e-4 * out1          # the loss is summed with fp32 atomics: not bitwise repeatable


@pytest.mark.parametrize("family", ["llama", "gemma"])
def test_generate_graph_replay_equals_eager_loop(family):
    """The decode step captured in a HIP graph (device-resident token / position / cache length) and replayed per token
    produces exactly the sequences of the eager loop, with and without an eos id (early stop, pad after eos)."""
    if family == "llama":
        zg, m = _load_generate()
        ids, mask = torch.from_numpy(zg["input_ids"]).cuda(), torch.from_numpy(zg["attention_mask"]).cuda()
        eos = int(zg["eos_token_id"])
    else:
        z, m = _load_gemma()
        m.eval()
        ids, mask = torch.from_numpy(z["gen_input_ids"]).cuda(), torch.from_numpy(z["gen_attention_mask"]).cuda()
        eos = int(z["gen_sequences"][1, ids.shape[1] + 3])
    for kw in (dict(), dict(eos_token_id=eos), dict(eos_token_id=[eos, 7])):
        eager = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=20, pad_token_id=299, use_graph=False, **kw)
        graph = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=20, pad_token_id=299, use_graph=True, **kw)
        assert eager.shape == graph.shape and torch.equal(eager, graph), kw
    one = m.generate(input_ids=ids[1:2], attention_mask=mask[1:2], max_new_tokens=20, pad_token_id=299, eos_token_id=eos)
    ref = m.generate(input_ids=ids[1:2], attention_mask=mask[1:2], max_new_tokens=20, pad_token_id=299, eos_token_id=eos, use_graph=False)
    assert torch.equal(one, ref)
    # a captured step is reused by later calls of the same batch size whose prompt + new tokens fit its buffers: other prompt lengths, other token counts
    m._gen_graphs.clear()
    for cut, new in ((0, 20), (3, 17), (1, 9), (0, 20)):
        kw = dict(input_ids=ids[:, cut:], attention_mask=mask[:, cut:], max_new_tokens=new, pad_token_id=299, eos_token_id=eos)
        assert torch.equal(m.generate(**kw), m.generate(use_graph=False, **kw)), (cut, new)
    assert len(m._gen_graphs) == 1


def test_data_parallel_gradients_two_ranks(tmp_path):
    """two ranks, each with its own rows of the batch, exchange gradients layer by layer during the
    backward pass (parallel.GradAllReduce); both must end with the mean of the two ranks' gradients, as
    DistributedDataParallel gives the reference (main.py:165)."""
    import subprocess
    import sys
    from ddp_worker import rank_rows
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = str(tmp_path / "grads")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(here, "ddp_worker.py"), out, "gloo"]
    res = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    want = None
    for rank in range(2):                                   # the same two backward passes, one after the other, no exchange
        z, m = _load()
        batch = {k: v[rank_rows(rank)] for k, v in _batch(z).items()}
        m(**batch).loss.backward()
        g = {n: p.grad.float() / 2 for n, p in m.named_parameters() if p.grad is not None}
        want = g if want is None else {n: want[n] + g[n] for n in g}
    for rank in range(2):
        got = np.load(out + f".rank{rank}.npz")
        assert int(got["__collectives__"]) >= 4                 # bucketed: several collectives per backward, far fewer than tensors
        assert set(got.files) - {"__collectives__"} == set(want)
        for n in want:
            a, b = torch.from_numpy(got[n]).cuda(), want[n]
            rel = (a - b).norm() / b.norm().clamp_min(1e-12)
            assert rel.item() < 2e-2, (rank, n, rel.item())


def test_rccl_single_rank_gradient_exchange(tmp_path):
    """The exchange over RCCL itself (backend "nccl"), as far as a one-GPU box can run it: one rank under torch.distributed.run
    initialises the communicator and sends every bucket of the flat gradient buffer through an asynchronous ncclAvg
    all-reduce issued between the backward kernels (which are launched through ctypes on torch's current stream).  The mean over
    one rank is the identity: the gradients must be those of a run without any exchange."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = str(tmp_path / "grads")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29523", os.path.join(here, "ddp_worker.py"), out, "nccl"]
    res = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    z, m = _load()
    m(**_batch(z)).loss.backward()
    got = np.load(out + ".rank0.npz")
    n_tensors = sum(1 for p in m.parameters() if p.grad is not None)
    assert int(got["__collectives__"]) == 6 < n_tensors          # two backward passes x (two layers + embedding/final norm)
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        a, b = torch.from_numpy(got[n]).cuda(), p.grad.float()
        rel = (a - b).norm() / b.norm().clamp_min(1e-12)
        assert rel.item() < 5e-3, (n, rel.item())                # same kernels; fp32 atomics reorder a few sums


def test_generate_long_prompt_takes_the_split_decode_attention(monkeypatch):
    """Past 512 cached keys the decode step splits a head's keys over workgroups (ecgb_attn_decode_split): the KV-cache
    path must still give the logits of the recompute-everything path (training kernels, no decode attention at all)."""
    from ecg_byte_amd import decoder_ops as ops
    zg, m = _load_generate()
    g = torch.Generator().manual_seed(3)
    B, S0, new = 2, 531, 6
    ids = torch.randint(3, 290, (B, S0), generator=g).cuda()
    mask = torch.ones(B, S0)
    mask[1, :9] = 0                                              # left padding on one row
    ids[1, :9] = 299
    mask = mask.cuda()
    m.decode_attn_one = False                                    # (round 4's four launches: what this test counts; the one-launch form has its own tests)
    calls = {"split": 0, "one": 0}
    real_split, real_one = ops.attn_decode_split, ops.attn_decode
    monkeypatch.setattr(ops, "attn_decode_split", lambda *a, **k: (calls.__setitem__("split", calls["split"] + 1), real_split(*a, **k))[1])
    monkeypatch.setattr(ops, "attn_decode", lambda *a, **k: (calls.__setitem__("one", calls["one"] + 1), real_one(*a, **k))[1])
    seq_c, lg_c = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=new, pad_token_id=299, use_cache=True, return_logits=True)
    assert calls["split"] == (new - 1) * m.cfg.num_hidden_layers and calls["one"] == 0
    seq_n, lg_n = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=new, pad_token_id=299, use_cache=False, return_logits=True)
    lg_c, lg_n = lg_c.float(), lg_n.float()
    tol = 1.5 * float(zg["ref_bf16_deviation"].max())
    for t in range(new):                                        # compare while the two greedy paths agree
        assert (lg_c[:, t] - lg_n[:, t]).abs().max().item() < tol, t
        if not torch.equal(seq_c[:, S0 + t], seq_n[:, S0 + t]):
            top2 = lg_n[:, t].topk(2, dim=-1).values
            assert (top2[:, 0] - top2[:, 1]).min().item() < tol  # a tie at bf16 resolution
            break


def test_generate_long_prompt_graph_replay_equals_eager_loop():
    """Caches of 512 rows and more: the replayed step runs the split decode attention with the length in device memory (ecgb_attn_decode_split_dyn), the
    eager loop the same kernels with the length as an argument and the same split count (by the caches' capacity): the same sequences, call after call."""
    zg, m = _load_generate()
    g = torch.Generator().manual_seed(5)
    B, S0 = 2, 540
    ids = torch.randint(3, 290, (B, S0), generator=g).cuda()
    mask = torch.ones(B, S0)
    mask[0, :17] = 0
    ids[0, :17] = 299
    mask = mask.cuda()
    seen = {}
    for one in (False, True):                                    # round 4's four launches; round 6's one-launch attention (the default)
        m.decode_attn_one = one
        m._gen_graphs = {}
        for cut, new in ((0, 30), (40, 70), (3, 12)):
            kw = dict(input_ids=ids[:, cut:], attention_mask=mask[:, cut:], max_new_tokens=new, pad_token_id=299)
            out = m.generate(use_graph=True, **kw)
            assert torch.equal(out, m.generate(use_graph=False, **kw)), (one, cut, new)
            assert torch.equal(seen.setdefault((cut, new), out), out)       # and both forms produce the same sequences
        st = next(iter(m._gen_graphs.values()))
        assert len(m._gen_graphs) == 1 and (st.scratch_one if one else st.scratch) is not None


def test_generate_sampling_follows_the_warped_distribution(tmp_path):
    """do_sample (what HF falls back to for a checkpoint whose generation_config.json says so -- the Llama-3.2 conversion writes
    do_sample=True, temperature 0.6, top_p 0.9): top_k = 1 is greedy; the empirical distribution of the first sampled token follows
    softmax(logits / T) restricted by top-k / top-p; generation_config.json is read by from_pretrained and sets the defaults."""
    import json
    zg, m = _load_generate()
    ids = torch.from_numpy(zg["input_ids"]).cuda()[:1]
    mask = torch.from_numpy(zg["attention_mask"]).cuda()[:1]
    greedy = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=6, pad_token_id=299)
    assert torch.equal(m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=6, pad_token_id=299, do_sample=True, top_k=1), greedy)
    _, lg = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=1, pad_token_id=299, return_logits=True)
    logits = lg[0, 0].float()
    T, K, P = 0.7, 12, 0.8
    sc = logits / T
    kth = sc.topk(K).values[-1]
    sc = sc.masked_fill(sc < kth, float("-inf"))
    srt, idx = sc.sort()
    rm = srt.softmax(-1).cumsum(-1) <= 1 - P
    rm[-1] = False
    sc[idx[rm]] = float("-inf")
    want = sc.softmax(-1)
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 4000
    rep_ids, rep_mask = ids.repeat(n // 8, 1), mask.repeat(n // 8, 1)
    draws = torch.cat([m.generate(input_ids=rep_ids, attention_mask=rep_mask, max_new_tokens=1, pad_token_id=299, do_sample=True,
                                  temperature=T, top_k=K, top_p=P, generator=g)[:, -1] for _ in range(8)])
    freq = torch.bincount(draws, minlength=want.numel()).float() / draws.numel()
    assert float(freq[want == 0].sum()) == 0.0                              # nothing outside the nucleus is ever drawn
    assert (freq - want).abs().max().item() < 0.04
    m.save_pretrained(str(tmp_path / "m"))
    with open(tmp_path / "m" / "generation_config.json", "w") as f:
        json.dump({"do_sample": True, "temperature": 0.6, "top_p": 0.9}, f)
    from ecg_byte_amd.decoder import HipCausalLM
    m2 = HipCausalLM.from_pretrained(str(tmp_path / "m"))
    assert m2.generation_config["do_sample"] is True
    a = m2.generate(input_ids=rep_ids[:64], attention_mask=rep_mask[:64], max_new_tokens=3, pad_token_id=299, generator=g)
    assert len({tuple(r.tolist()) for r in a[:, -3:]}) > 1                  # sampled by default now: the rows differ


@pytest.mark.parametrize("S", [100, 1004 % 256 + 64])
def test_sequence_length_not_a_multiple_of_64(S):
    """The reference's default --pad_to_max 1000 gives rows of 1004 tokens (data_loader.py:123): the training forward pads such rows on
    the LEFT with masked, unlabelled positions up to the GEMM K-step; loss and gradients must be those of the unpadded rows (fp32 oracle),
    and the eval loss the same number."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    from oracle import llama_ref as R
    cfgd = dict(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6)
    params = R.random_params(cfgd, seed=11, device="cuda", std=0.05)
    m = HipCausalLM(DecoderConfig(vocab_size=515, hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=4,
                                  num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None, pad_token_id=514))
    m.load_state_dict(params)
    B = 3
    g = torch.Generator(device="cuda").manual_seed(S)
    ids = torch.randint(0, 514, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda"); mask[1, :23] = 0; ids[1, :23] = 514
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device="cuda"); labels[:, -17:] = ids[:, -17:]
    ref_p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = R.llama_loss(ref_p, cfgd, ids, mask, labels, pos, R.llama3_inv_freq(64, 10000.0, None).cuda())
    ref.backward()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item()
    for i in range(2):
        want = torch.cat([ref_p[f"model.layers.{i}.self_attn.{n}_proj.weight"].grad for n in "qkv"], 0)
        assert ((m.wqkv[i].grad.float() - want).norm() / want.norm()).item() < 3e-2
        want = ref_p[f"model.layers.{i}.mlp.down_proj.weight"].grad
        assert ((m.wdown[i].grad.float() - want).norm() / want.norm()).item() < 3e-2
    want = ref_p["model.embed_tokens.weight"].grad
    assert ((m.embed.grad[:515].float() - want).norm() / want.norm()).item() < 3e-2
    with torch.no_grad():
        ev = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos).loss.item()
    assert abs(ev - ref.item()) <= 1e-2 * ref.item()


@pytest.mark.parametrize("family", ["llama", "gemma"])
def test_small_width_step_is_the_same_bits_twice(family):
    """Widths that take the generic RMSNorm backward (one wave per workgroup, per-workgroup partial rows added in order), split-K weight
    gradients over slabs, the sorted embedding scatter: two models from one seed, one step each, identical loss / gradients / weights."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM

    def step():
        cfg = DecoderConfig(vocab_size=700, hidden_size=192, intermediate_size=448, num_hidden_layers=2, num_attention_heads=3,
                            num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=None, pad_token_id=699,
                            model_type=family)
        m = HipCausalLM(cfg, seed=9)
        opt = m.make_optimizer(warmup=10)
        g = torch.Generator(device="cuda").manual_seed(4)
        ids = torch.randint(0, 699, (3, 192), device="cuda", generator=g)
        mask = torch.ones(3, 192, device="cuda"); mask[0, :50] = 0; ids[0, :50] = 699
        pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
        labels = torch.full((3, 192), -100, device="cuda"); labels[:, -30:] = ids[:, -30:]
        out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
        out.loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step_and_update_lr()
        return out.loss.detach().clone(), grads, {n: p.data.clone() for n, p in m.named_parameters()}
    l1, g1, p1 = step()
    l2, g2, p2 = step()
    assert torch.equal(l1, l2) and len(g1) > 0
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in p1:
        assert torch.equal(p1[k], p2[k]), k
