import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import os, sys
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
import test_gpu_decoder_model as T
z, m = T._load_gemma()
out = m(**T._batch(z)); print("loss", out.loss.item(), float(z["loss_fp32"]))
out.loss.backward()
c = m.cfg
D, Hq, Hkv, I = c.head_dim, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size
grads = {"model.embed_tokens.weight": m.embed.grad[: c.vocab_size], "model.norm.weight": m.norm.grad}
for i in range(c.num_hidden_layers):
    p = f"model.layers.{i}."
    g = m.wqkv[i].grad
    grads[p + "self_attn.q_proj.weight"] = g[: Hq * D]; grads[p + "self_attn.k_proj.weight"] = g[Hq * D: Hq * D + Hkv * D]; grads[p + "self_attn.v_proj.weight"] = g[Hq * D + Hkv * D:]
    grads[p + "self_attn.o_proj.weight"] = m.wo[i].grad; grads[p + "mlp.gate_proj.weight"] = m.wgu[i].grad[:I]; grads[p + "mlp.up_proj.weight"] = m.wgu[i].grad[I:]
    grads[p + "mlp.down_proj.weight"] = m.wdown[i].grad; grads[p + "input_layernorm.weight"] = m.ln1[i].grad; grads[p + "post_attention_layernorm.weight"] = m.ln2[i].grad
for name, g in grads.items():
    ref = torch.from_numpy(z["g:" + name]).cuda(); g = g.float()
    print("%-50s rel %.4f cos %.5f  |ref| %.3g" % (name, ((g - ref).norm() / ref.norm()).item(), torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0).item(), ref.norm().item()))
