"""Mirror of ecg_byte/models/llm.py:7-37: the wrapper main.py puts around the causal LM.  `llm` is a
`decoder.HipCausalLM` (or anything with the same surface)."""
from __future__ import annotations

import torch.nn as nn


class LLM(nn.Module):
    def __init__(self, llm, args):
        super().__init__()
        self.args = args
        self.llm = llm
        self.output_attentions = bool(getattr(args, "interpret", False))

    def forward(self, batch):
        """llm.py:17-24"""
        dev = self.llm.device
        return self.llm(input_ids=batch["tokenized_signal"].to(dev), attention_mask=batch["attn_mask"].to(dev),
                        labels=batch["quantized_signal_ids_input"].to(dev), position_ids=batch["position_ids"].to(dev),
                        output_attentions=self.output_attentions)

    def generate(self, batch, tokenizer):
        """llm.py:26-37: greedy continuation of the prompt, decoded without the prompt, first sequence only."""
        dev = self.llm.device
        input_len = batch["tokenized_signal"].shape[1]
        generated_ids = self.llm.generate(input_ids=batch["tokenized_signal"].to(dev), attention_mask=batch["attn_mask"].to(dev),
                                          max_new_tokens=128, pad_token_id=tokenizer.pad_token_id,
                                          eos_token_id=tokenizer.eos_token_id, use_cache=True)
        return tokenizer.batch_decode(generated_ids[:, input_len:], skip_special_tokens=True,
                                      clean_up_tokenization_spaces=False)[0]

    # checkpoints are `{'model': model.state_dict(), 'epoch': epoch}` with this module's prefix (main.py:299-306)
    def state_dict(self, *a, **k):
        return {"llm." + n: t for n, t in self.llm.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        return self.llm.load_state_dict({n[4:] if n.startswith("llm.") else n: t for n, t in sd.items()}, strict=strict)
