"""GPU tests of the GPT-2 block (ecg_byte_amd/gpt2.py, BASELINE config C1's model) against goldens generated from the VENDORED
transformers GPT2LMHeadModel (tests/golden/make_decoder_golden_gpt2.py) and, at GPT-2-small dimensions, against the fp32 oracle
(oracle/gpt2_ref.py, itself pinned to those goldens).  Tolerances as for the Llama block: loss within 1e-2 relative of the fp32 run,
every gradient within 3e-2 in relative Frobenius norm."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _load():
    z = np.load(os.path.join(GOLDEN, "decoder_gpt2_tiny.npz"))
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM
    cfg = GPT2Config(vocab_size=300, n_positions=128, n_embd=128, n_layer=2, n_head=2, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                     initializer_range=0.05, pad_token_id=299)
    m = HipGPT2LM(cfg)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    return z, m


def _batch(z):
    return dict(input_ids=torch.from_numpy(z["input_ids"]).cuda(), attention_mask=torch.from_numpy(z["attention_mask"]).cuda(),
                labels=torch.from_numpy(z["labels"]).cuda(), position_ids=torch.from_numpy(z["position_ids"]).cuda())


def _grads_hf(m):
    """HF-named gradients of a HipGPT2LM (Conv1D weights transposed back to [in, out])."""
    L = m.cfg.n_layer
    out = {"transformer.wte.weight": m.embed.grad[: m.cfg.vocab_size].float(), "transformer.wpe.weight": m.wpe.grad.float(),
           "transformer.ln_f.weight": m.norm.grad.float(), "transformer.ln_f.bias": m.norm_b.grad.float()}
    for i in range(L):
        p = f"transformer.h.{i}."
        for name, prm, tr in (("ln_1.weight", m.ln1, 0), ("ln_1.bias", m.ln1_b, 0), ("attn.c_attn.weight", m.wqkv, 1), ("attn.c_attn.bias", m.bqkv, 0),
                              ("attn.c_proj.weight", m.wo, 1), ("attn.c_proj.bias", m.bo, 0), ("ln_2.weight", m.ln2, 0), ("ln_2.bias", m.ln2_b, 0),
                              ("mlp.c_fc.weight", m.wfc, 1), ("mlp.c_fc.bias", m.bfc, 0), ("mlp.c_proj.weight", m.wproj, 1), ("mlp.c_proj.bias", m.bproj, 0)):
            g = prm[i].grad.float()
            out[p + name] = g.t() if tr else g
    return out


@pytest.mark.parametrize("full_logits", [False, True])
def test_gpt2_loss_and_gradients_vs_vendored_transformers(full_logits):
    z, m = _load()
    m.full_logits = full_logits
    out = m(**_batch(z))
    ref = float(z["loss_fp32"])
    assert abs(out.loss.item() - ref) <= 1e-2 * ref, (out.loss.item(), ref)
    out.loss.backward()
    for name, g in _grads_hf(m).items():
        want = torch.from_numpy(z["g:" + name]).cuda()
        rel = ((g - want).norm() / want.norm().clamp_min(1e-12)).item()
        assert rel < 3e-2, (name, rel)


def test_gpt2_logits_and_eval_loss():
    z, m = _load()
    m.eval()
    b = _batch(z)
    logits = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], position_ids=b["position_ids"]).logits
    want = torch.from_numpy(z["logits_fp32"]).cuda()
    assert (logits[:, -3:] - want).abs().max().item() < 0.06 * want.abs().max().item()
    with torch.no_grad():
        loss = m(**b).loss.item()
    assert abs(loss - float(z["loss_fp32"])) <= 1e-2 * float(z["loss_fp32"])


def test_gpt2_state_dict_roundtrip_and_pretrained_directory(tmp_path):
    z, m = _load()
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("w:"):
            assert torch.equal(sd[k[2:]].cpu().float(), torch.from_numpy(z[k])), k     # the golden weights are bf16-representable
    from ecg_byte_amd.decoder import HipCausalLM
    m.save_pretrained(str(tmp_path / "g"))
    assert json.load(open(tmp_path / "g" / "config.json"))["model_type"] == "gpt2"
    m2 = HipCausalLM.from_pretrained(str(tmp_path / "g"))                             # dispatches on model_type
    a, b = m.state_dict(), m2.state_dict()
    assert type(m2).__name__ == "HipGPT2LM" and a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("use_cache", [True, False], ids=["kv-cache", "recompute"])
def test_gpt2_generate_greedy_vs_vendored_transformers(use_cache):
    from test_gpu_decoder_model import _check_greedy
    z, m = _load()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m.load_state_dict({k: (v * 4.0 if k.endswith(("c_attn.weight", "c_proj.weight", "c_fc.weight")) else v) for k, v in sd.items()})
    m.eval()
    ids = torch.from_numpy(z["gen_input_ids"]).cuda()
    mask = torch.from_numpy(z["gen_attention_mask"]).cuda()
    S0 = ids.shape[1]
    seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=16, pad_token_id=299, use_cache=use_cache, return_logits=True)
    exact = _check_greedy(seq.cpu().numpy(), logits.cpu().numpy(), z["gen_sequences"], z["gen_scores"], S0,
                          1.5 * float(z["gen_ref_bf16_deviation"].max()))
    assert exact >= 24, exact


def test_gpt2_residual_dropout_replayed_in_backward():
    """embd / resid dropout in training mode: finite loss, gradients differ from the dropout-free run, and a second backward of the
    same forward state is impossible (state consumed) -- the masks are counter-based and replayed, never stored."""
    z, m = _load()
    m.cfg.resid_pdrop, m.cfg.embd_pdrop = 0.1, 0.1
    m.train()
    out = m(**_batch(z))
    out.loss.backward()
    g1 = m.wfc[0].grad.float().clone()
    assert torch.isfinite(out.loss) and torch.isfinite(g1).all()
    m.eval()
    for p in m.parameters():
        p.grad = None
    out2 = m(**_batch(z))
    out2.loss.backward()
    assert ((m.wfc[0].grad.float() - g1).norm() / g1.norm()).item() > 1e-2


def test_gpt2_attention_dropout_matches_a_torch_restatement_with_the_same_masks():
    """attn_pdrop > 0 in training mode (GPT2Config's default): dropout on the attention probabilities (modeling_gpt2.py:214) through the
    materialised-scores path.  The masks the kernels drew are read back from the saved (P, dropped P) pairs and fed to the fp32 oracle
    block with autograd: loss within 1e-2, every gradient within 4e-2 (bf16 probabilities in the product path)."""
    from oracle import gpt2_ref as G
    import torch.nn.functional as F
    z, m = _load()
    m.cfg.attn_pdrop = 0.2
    m.train()
    b = _batch(z)
    out = m(**b)
    saved = m._saved[0]
    keeps = []
    for layer in saved:
        (P, Pd), (p, seed) = layer[5]
        keeps.append(((Pd != 0) | (P == 0)).view(3, 2, 64, 64))
        rate = 1 - ((Pd != 0).sum() / (P != 0).sum()).item()
        assert abs(rate - 0.2) < 0.03, rate
    out.loss.backward()
    params = {k[2:]: torch.from_numpy(z[k]).cuda().requires_grad_(True) for k in z.files if k.startswith("w:")}
    ids, mask, pos = b["input_ids"], b["attention_mask"], b["position_ids"]
    H, nh, D, Bn, S = 128, 2, 64, 3, 64
    x = params["transformer.wte.weight"][ids] + params["transformer.wpe.weight"][pos]
    vis = torch.tril(torch.ones(S, S, dtype=torch.bool, device="cuda"))[None, None] & (mask[:, None, None, :] != 0)
    for i in range(2):
        pre = f"transformer.h.{i}."
        h = F.layer_norm(x, (H,), params[pre + "ln_1.weight"], params[pre + "ln_1.bias"], 1e-5)
        q, k, v = (h @ params[pre + "attn.c_attn.weight"] + params[pre + "attn.c_attn.bias"]).split(H, dim=2)
        q, k, v = (t.view(Bn, S, nh, D).transpose(1, 2) for t in (q, k, v))
        sc = (q @ k.transpose(-1, -2)) / 8.0
        pr = torch.nan_to_num(torch.softmax(sc.masked_fill(~vis, float("-inf")), -1), nan=0.0)
        pr = pr * keeps[i].float() / (1 - 0.2)
        a = (pr @ v).transpose(1, 2).reshape(Bn, S, H)
        x = x + a @ params[pre + "attn.c_proj.weight"] + params[pre + "attn.c_proj.bias"]
        h = F.layer_norm(x, (H,), params[pre + "ln_2.weight"], params[pre + "ln_2.bias"], 1e-5)
        x = x + G.gelu_new(h @ params[pre + "mlp.c_fc.weight"] + params[pre + "mlp.c_fc.bias"]) @ params[pre + "mlp.c_proj.weight"] + params[pre + "mlp.c_proj.bias"]
    x = F.layer_norm(x, (H,), params["transformer.ln_f.weight"], params["transformer.ln_f.bias"], 1e-5)
    logits = x @ params["transformer.wte.weight"].T
    ref = F.cross_entropy(logits[:, :-1].reshape(-1, 300), b["labels"][:, 1:].reshape(-1), ignore_index=-100)
    ref.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item(), (out.loss.item(), ref.item())
    for name, g in _grads_hf(m).items():
        want = params[name].grad
        rel = ((g - want).norm() / want.norm().clamp_min(1e-12)).item()
        assert rel < 4e-2, (name, rel)


def test_gpt2_small_dims_two_layers_vs_fp32_oracle():
    """C1's model at its real width: GPT-2-small dims (768 hidden, 12 heads, 3072 MLP, vocab 50 257 + 256 + 1 000 + 3, positions 1024),
    two layers, B 4 (the batch of BASELINE config C1), S 1024, left-padded, -100 labels; oracle/gpt2_ref.py in fp32 on the same GPU."""
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM
    from oracle import gpt2_ref as G
    V = 50257 + 256 + 1000 + 3
    cfgd = dict(vocab_size=V, n_positions=1024, n_embd=768, n_layer=2, n_head=12)
    params = G.random_params(cfgd, seed=5, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(6)
    for k in params:                                                          # non-trivial biases and norms
        if k.endswith(".bias") or "ln_" in k:
            params[k] = (params[k] + 0.05 * torch.randn(params[k].shape, device="cuda", generator=g)).to(torch.bfloat16).float()
    B, S = 4, 1024
    ids = torch.randint(0, V - 1, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda")
    for b, n in enumerate([300, 0, 37, 777]):
        mask[b, :n] = 0
        ids[b, :n] = V - 1
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
    pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device="cuda")
    labels[:, -40:] = ids[:, -40:]
    ref_p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = G.gpt2_loss(ref_p, cfgd, ids, mask, labels, pos)
    ref.backward()
    m = HipGPT2LM(GPT2Config(vocab_size=V, n_layer=2, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, pad_token_id=V - 1))
    m.load_state_dict(params)
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    assert abs(out.loss.item() - ref.item()) <= 1e-2 * ref.item(), (out.loss.item(), ref.item())
    for name, gq in _grads_hf(m).items():
        want = ref_p[name].grad
        rel = ((gq - want).norm() / want.norm().clamp_min(1e-12)).item()
        assert rel < 3e-2, (name, rel)


def test_gpt2_step_is_the_same_bits_twice():
    """GPT-2-small widths, two layers, dropout on (counter-based masks: the same seeds draw the same masks): two models from one seed take
    one step each -- loss, every gradient (LayerNorm weights / biases and the projection biases are summed over per-workgroup partial rows in
    order, the embeddings by the sorted scatter) and every updated weight identical."""
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM

    def step():
        cfg = GPT2Config(vocab_size=4099, n_layer=2)
        m = HipGPT2LM(cfg, seed=5)
        m.train()
        opt = m.make_optimizer()
        g = torch.Generator(device="cuda").manual_seed(3)
        ids = torch.randint(0, 4098, (4, 1024), device="cuda", generator=g)
        mask = torch.ones(4, 1024, device="cuda"); mask[1, :200] = 0; mask[3, :17] = 0
        pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long()
        labels = torch.full((4, 1024), -100, device="cuda"); labels[:, -50:] = ids[:, -50:]
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
