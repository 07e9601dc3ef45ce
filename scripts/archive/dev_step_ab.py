"""Dev-only: the C3 training step (full fine-tune and LoRA) with the persistent GEMM tile loops on / off (set_gemm_tile(0) / (259)), alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecg_byte_amd import decoder_ops as ops
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
B, S = 32, 1024
V = 128256 + 256 + 3500 + 3
for lora in (False, True):
    cfg = DecoderConfig.llama_3_2_1b(vocab_size=V, pad_token_id=V - 1)
    m = HipCausalLM(cfg)
    if lora: m.enable_lora(r=16, alpha=32, dropout=0.05)
    m.train()
    opt = m.make_optimizer()
    g = torch.Generator(device="cuda").manual_seed(0)
    ids = torch.randint(1000, 100000, (B, S), device="cuda", generator=g)
    mask = torch.ones(B, S, device="cuda"); mask[:, :100] = 0; ids[:, :100] = cfg.pad_token_id
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device="cuda"); labels[:, -40:] = ids[:, -40:]
    def step():
        opt.zero_grad()
        out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
        out.loss.backward()
        opt.step_and_update_lr()
    for _ in range(3): step()
    res = {0: [], 259: []}
    for rnd in range(3):
        for code in (259, 0):
            ops.set_gemm_tile(code)
            step(); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(6): step()
            torch.cuda.synchronize(); res[code].append((time.perf_counter() - t) / 6 * 1e3)
    ops.set_gemm_tile(0)
    print("lora" if lora else "full", "one tile per workgroup:", [round(x, 1) for x in res[259]], " persistent:", [round(x, 1) for x in res[0]])
    del m, opt
    torch.cuda.empty_cache()
