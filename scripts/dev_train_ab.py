"""Dev-only: the C3 train step (Llama-3.2-1B dims, seq 1024, batch 32, synthetic ids) with a switch flipped between rounds in ONE process -- boxes differ by 2 %,
an A/B across gpurun calls cannot see 1 ms.  Usage: dev_train_ab.py [w4|rope_bwd|rope_fwd|r4sched|r4dispatch|r4all|lora_dx] [lora]
(r4sched: the four-wave kernel's round-3 / round-4 K-tile schedule; r4dispatch: its dispatch threshold 256 / 128 K-tiles per workgroup; r4all: both; lora_dx: the down site's product + adapter share + GLU backward in one launch, LoRA only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
what = sys.argv[1] if len(sys.argv) > 1 else "w4"
lora = len(sys.argv) > 2
V = 128256 + 4000 + 259
cfg = DecoderConfig.llama_3_2_1b(vocab_size=V, pad_token_id=V - 1)
m = HipCausalLM(cfg, device="cuda", seed=0)
if lora:
    m.enable_lora(r=16, alpha=32, dropout=0.05)
opt = m.make_optimizer()
g = torch.Generator(device="cuda").manual_seed(0)
B, S = 32, 1024
ids = torch.randint(1000, 100000, (B, S), device="cuda", generator=g)
mask = torch.ones(B, S, device="cuda")
labels = torch.full((B, S), -100, device="cuda", dtype=torch.long)
labels[:, -24:] = ids[:, -24:]
pos = torch.arange(S, device="cuda").repeat(B, 1)
def step():
    opt.zero_grad()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
    out.loss.backward()
    opt.step_and_update_lr()
def r4sched(on): ops.set_gemm_w4_sched(1 if on else 0)
def r4dispatch(on): ops.set_gemm_w4_min_ktiles(128 if on else 256)
def r4all(on): r4sched(on); r4dispatch(on)
switch = {"w4": ops.set_gemm_w4, "rope_bwd": ops.set_attn_bwd_rope_fusion, "r4sched": r4sched, "r4dispatch": r4dispatch, "r4all": r4all, "lora_dx": ops.set_fuse_lora_dx_glu, "rope_fwd": ops.set_gemm_rope_fusion}[what]
for _ in range(3): step()
res = {False: [], True: []}
for rnd in range(4):
    for on in (False, True):
        switch(on)
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): step()
        e1.record(); torch.cuda.synchronize()
        res[on].append(e0.elapsed_time(e1) / 5)
switch(True)
print(f"{'LoRA' if lora else 'full fine-tune'} step, {what} off: {min(res[False]):.2f} ms (median {sorted(res[False])[2]:.2f}), on: {min(res[True]):.2f} ms (median {sorted(res[True])[2]:.2f})")
