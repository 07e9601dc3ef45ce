"""Dev-only: greedy generate at Gemma-2B dims with adapters (the C5 shape: 600-token prompt + 128 new tokens, batch 1, replayed graph), a model switch flipped in one process:
python scripts/dev_gen_lora_ab.py decode_lora_one [decode_attn_one ...] -- every named class attribute False then True, best of three each, and the sequences compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
V = 256000 + 256 + 3500 + 3
m = HipCausalLM(DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1))
m.enable_lora(16, 32, 0.05)
for n_, p_ in m.named_parameters():
    if "lora_B" in n_ or n_.endswith(".B"):
        p_.data.normal_(0, 0.01)
m.eval()
p = torch.randint(1000, 100000, (1, 600), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
pm = torch.ones_like(p, dtype=torch.float32)
for attr in sys.argv[1:] or ["decode_lora_one"]:
    seqs = {}
    for rnd in range(2):
        for on in (False, True):
            setattr(m, attr, on)
            best = None
            for it in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                seq = m.generate(input_ids=p, attention_mask=pm, max_new_tokens=128, pad_token_id=V - 1)
                torch.cuda.synchronize(); dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            seqs[on] = seq
            print(f"{attr} = {on}: {1e3 * best:.1f} ms ({128 / best:.1f} tokens/s)", flush=True)
    print(f"{attr}: same tokens {torch.equal(seqs[False], seqs[True])}")
    setattr(m, attr, True)
