#!/usr/bin/env python3
"""Regenerates tests/golden/decoder_generate_tiny.npz from the VENDORED transformers under /root/reference
(build container only): greedy `generate` (do_sample=False, use_cache=True, as ecg_byte/models/llm.py:26-37
calls it) of the tiny Llama whose weights are those of decoder_llama_tiny.npz with every projection scaled by 4, on a left-padded prompt
batch whose length is not a multiple of 64.  eos_token_id is chosen as a token one sequence emits at
step 5 so the finished-sequence padding and the early-stop rule are exercised.  Stored: prompts, mask,
generated ids, the fp32 scores of every step, and the deviation of the
reference's own bf16 run from them."""
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
    z = np.load(os.path.join(HERE, "decoder_llama_tiny.npz"))
    cfg = LlamaConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, rms_norm_eps=1e-5,
                      rope_theta=500000.0, tie_word_embeddings=True, pad_token_id=299, initializer_range=0.05,
                      rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                    "original_max_position_embeddings": 32, "rope_type": "llama3"})
    m = LlamaForCausalLM(cfg).eval()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    # the training fixture's weights make a tied-embedding model repeat its last token; scale the projections up so the
    # layers dominate the residual stream and the greedy path wanders (still bf16-representable: powers of two)
    sd = {k: (v * 4.0 if "proj" in k else v) for k, v in sd.items()}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(11)
    B, S0, NEW = 3, 45, 24
    ids = torch.randint(0, 299, (B, S0), generator=g)
    mask = torch.ones(B, S0, dtype=torch.long)
    mask[0, :9] = 0; ids[0, :9] = 299
    mask[2, :31] = 0; ids[2, :31] = 299
    kw = dict(input_ids=ids, attention_mask=mask, max_new_tokens=NEW, pad_token_id=299, use_cache=True, do_sample=False,
              output_scores=True, return_dict_in_generate=True)
    free = m.generate(**kw)
    eos = int(free.sequences[1, S0 + 5])
    assert eos not in free.sequences[1, S0:S0 + 5].tolist()
    out = m.generate(eos_token_id=eos, **kw)
    nocache = m.generate(eos_token_id=eos, **{**kw, "use_cache": False})
    assert torch.equal(out.sequences, nocache.sequences)
    scores = torch.stack(out.scores, 1).numpy()
    # how far the reference's own bf16 run is from its fp32 run, teacher-forced on the fp32 greedy path: the yardstick
    # for the tolerance of the bf16 HIP path
    mb = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    mb.load_state_dict({k: v.to(torch.bfloat16) for k, v in m.state_dict().items()})
    full_mask = torch.cat([mask, torch.ones(B, NEW, dtype=torch.long)], 1)
    pos = (full_mask.cumsum(-1) - 1).masked_fill(full_mask == 0, 1)
    with torch.no_grad():
        lb = mb(input_ids=free.sequences, attention_mask=full_mask, position_ids=pos).logits.float()[:, S0 - 1:-1]
    bf16_dev = (lb - torch.stack(free.scores, 1)).abs().amax(-1).numpy()
    print("reference bf16-vs-fp32 logit deviation: max %.3f mean %.3f" % (bf16_dev.max(), bf16_dev.mean()))
    np.savez_compressed(os.path.join(HERE, "decoder_generate_tiny.npz"), input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        eos_token_id=np.int64(eos), pad_token_id=np.int64(299), max_new_tokens=np.int64(NEW),
                        sequences=out.sequences.numpy(), scores=scores.astype(np.float32),
                        sequences_no_eos=free.sequences.numpy(), scores_no_eos=torch.stack(free.scores, 1).numpy().astype(np.float32),
                        ref_bf16_deviation=bf16_dev.astype(np.float32))
    print("eos", eos, "generated", out.sequences.shape, "free", free.sequences.shape)
    print(out.sequences[:, S0:])


if __name__ == "__main__":
    main()
