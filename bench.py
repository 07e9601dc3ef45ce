#!/usr/bin/env python3
"""bench.py -- ECG-tokens/sec of the quantise + BPE-encode hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): synthetic PTB-XL-shaped records, 500 Hz x 10 s,
(B, 12, 5000) float64 resident in HBM, the 4000-merge tokenizer of tests/golden/tokenizer_c2.pkl.
One step = one pass of the hot path (normalize_all -> symbol stream -> greedy longest-match
token ids) over the whole batch.  Each rank owns B records (weak scaling, no collective on
the data path: records are independent).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = token ids produced per second by the whole job
(inputs already resident in HBM).  `roofline` prices the timed kernels against HBM with the
algorithmic bytes of SURVEY.md §8d: 8*12*L read + 4*T written per record.  `cpu_baseline` times
the CPU oracle (a C port of the reference's Rust encoder + numpy quantiser) on this box's host
cores over a bounded sample of the same records.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def compact(o, sig=6):
    """Floats to `sig` significant digits (the line must fit the 8 KB the driver retains; no figure in it is good to more than four)."""
    if isinstance(o, float):
        return float(f"{o:.{sig}g}") if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: compact(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, sig) for v in o]
    return o


def with_top_level_scalars(out):
    """The results BASELINE.json's metric names, as scalars at the top level of the line (a record that keeps only the top level still carries them): the encode
    figure is `value`; the train-step figures of the C3 legs, the full-size parity gate, generate and the tokenizer trainer follow."""
    def g(*path):
        d = out
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    top = {}
    for k, v in out.items():
        top[k] = v
        if k == "data":
            top["encode_roofline_frac"] = g("roofline", "frac")
            top["train_samples_per_s"] = g("train", "value")
            top["train_ms_per_step"] = g("train", "ms_per_step")
            top["train_roofline_frac"] = g("train", "roofline", "frac")
            top["train_lora_samples_per_s"] = g("train", "lora_r16", "value")
            top["train_lora_ms_per_step"] = g("train", "lora_r16", "ms_per_step")
            top["train_lora_roofline_frac"] = g("train", "lora_r16", "roofline", "frac")
            top["train_full_logits_ms_per_step"] = g("train", "full_logits", "ms_per_step")
            top["train_full_logits_roofline_frac"] = g("train", "full_logits", "roofline", "frac")
            top["train_loss_rel_err_vs_fp32_oracle"] = g("train", "cpu_baseline", "parity_gate", "loss_rel_err_vs_fp32_oracle")
            top["train_lora_loss_rel_err_vs_fp32_oracle"] = g("train", "lora_r16", "cpu_baseline", "parity_gate", "loss_rel_err_vs_fp32_oracle")
            top["c5_loss_rel_err_vs_fp32_oracle"] = g("c5", "train", "cpu_baseline", "parity_gate", "loss_rel_err_vs_fp32_oracle")
            top["c1_loss_rel_err_vs_fp32_oracle"] = g("c1", "parity_gate", "loss_rel_err_vs_fp32_oracle")
            top["train_lora_ms_per_step_from_disk"] = g("train", "lora_r16", "loader", "ms_per_step_from_disk")
            top["train_cpu_samples_per_s"] = g("train", "cpu_baseline", "value")
            top["c5_lora_ms_per_step"] = g("c5", "train", "ms_per_step")
            top["c5_generate_tokens_per_s"] = g("c5", "generate", "tokens_per_s")
            top["c5_decode_tokens_per_s"] = g("c5", "generate", "decode_tokens_per_s")
            top["c5_decode_roofline_frac"] = g("c5", "generate", "roofline", "frac")
            top["c5_decode_tokens_per_s_merged_adapters"] = g("c5", "generate", "merged_adapters", "decode_tokens_per_s")
            top["trainer_seconds"] = g("trainer", "seconds")
            top["trainer_roofline_frac"] = g("trainer", "frac_of_hbm_peak")
            top["preprocess_ms"] = g("preprocess", "ms")
    return {k: v for k, v in top.items() if v is not None or k == "vs_baseline"}


def _latest_profile(name):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    return files[-1] if files else None


def measured_traffic_bytes():
    """(HBM bytes per launch, source) from the committed rocprofv3 PMC passes of this same workload (profiles/r*/hbm_pmc.json;
    FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for 16 B/lane coalesced reads).  The counters need their own
    rocprofv3 passes, so the figure is NOT measured by the run that prints it: `traffic_source` names the profile.  (None, None) if
    no profile is committed."""
    f = _latest_profile("hbm_pmc.json")
    if not f:
        return None, None
    with open(f) as fh:
        d = json.load(fh)
    d = d.get("counters", d)                  # (scripts/pmc_summary.py's own layout, or the committed one with the counters at the top level)
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None, None
    return (2.0 * d["FETCH_SIZE"]["per_launch_mean"] + d["WRITE_SIZE"]["per_launch_mean"]) * 1024.0, os.path.relpath(f, ROOT)


def _gen_chunk(args):
    from ecg_byte_amd import synth
    start, count, L, seed = args
    return synth.synth_ecg(count, L, seed=seed, start=start)


def make_signals(B, L, seed, start, workers):
    """Records start..start+B-1 of the seeded generator, built with a small process pool."""
    from ecg_byte_amd import synth
    if workers <= 1 or B < 64:
        return synth.synth_ecg(B, L, seed=seed, start=start)
    import multiprocessing as mp
    step = max(16, B // (workers * 4))
    jobs = [(start + s, min(step, B - s), L, seed) for s in range(0, B, step)]
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_gen_chunk, jobs)
    return np.concatenate(parts, axis=0)


def cpu_baseline(merges, pc, L, seed, budget_s=12.0):
    """Times the CPU oracle on this host: 1 core, one record at a time, the way the reference's
    DataLoader(num_workers=0) runs it.  `value` = reference-faithful variant (trie rebuilt on
    every call, lib.rs:153-161); the build-once variant is reported beside it."""
    from oracle import oracle as O
    from ecg_byte_amd import synth
    p1, p99 = pc["percentile_1"], pc["percentile_99"]
    x = synth.synth_ecg(64, L, seed=seed)
    trie = O.Trie(merges)
    # build-once variant
    t0 = time.perf_counter(); toks = 0; n_once = 0
    while time.perf_counter() - t0 < budget_s / 3:
        toks += trie.quantize_encode(x[n_once % 64], p1, p99).size
        n_once += 1
    dt_once = time.perf_counter() - t0
    # reference-faithful: quantise + rebuild trie + encode per record
    t0 = time.perf_counter(); toks_f = 0; n_f = 0
    while time.perf_counter() - t0 < budget_s * 2 / 3:
        sym = O.quantize(x[n_f % 64], p1, p99)
        toks_f += len(O.encode_text(O.symbols_to_text(sym), merges))
        n_f += 1
    dt_f = time.perf_counter() - t0
    # the quantiser alone: Python-level per-sample loop as the reference runs it, and the C restatement
    t0 = time.perf_counter(); n_py = 0
    while time.perf_counter() - t0 < 1.0:
        O.quantize_python_style(x[n_py % 64], p1, p99)
        n_py += 1
    dt_py = time.perf_counter() - t0
    t0 = time.perf_counter(); n_c = 0
    while time.perf_counter() - t0 < 0.5:
        O.quantize(x[n_c % 64], p1, p99)
        n_c += 1
    dt_c = time.perf_counter() - t0
    # all host cores, one record per thread at a time; the C calls release the GIL
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(os.cpu_count() or 1, 64))
    per_thread = max(1, int(n_once / max(dt_once, 1e-9) * 2.5))                  # about 2.5 s of work per thread

    def work(k):
        tr = O.Trie(merges)
        return sum(tr.quantize_encode(x[(k + i) % 64], p1, p99).size for i in range(per_thread))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as pool:
        toks_all = sum(pool.map(work, range(n_thr)))
    dt_all = time.perf_counter() - t0
    return {
        "value": toks_f / dt_f, "unit": "tokens/s", "cores": 1, "kind": "port",
        "value_all_cores_trie_built_once": toks_all / dt_all, "threads_all_cores": n_thr,
        "sample": f"{n_f} records of 12x{L} in {dt_f:.1f} s, trie rebuilt per call (as the reference); oracle/ecgb_oracle.c",
        "records_per_s": n_f / dt_f,
        "value_trie_built_once": toks / dt_once,
        "quantiser_python_style_symbols_per_s": n_py * 12 * L / dt_py,      # np.vectorize + join, one core (tokenizer_utils.py:14-19)
        "quantiser_c_symbols_per_s": n_c * 12 * L / dt_c,
        "host_cpus": os.cpu_count(),
    }


MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s


LLAMA3_ROPE = (500000.0, {"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192})


def _cpu_train_once(cfg_kw, S, threads, dtype, with_optimizer, base_params, batch=None, lora_scale=None, rope=LLAMA3_ROPE):
    """One forward + backward of oracle/llama_ref.py on the host.  `batch` = (ids, mask, labels, positions) of ONE sample, or None for a synthetic one.
    lora_scale: the frozen-base LoRA step (only the `.lora_A / .lora_B` entries of `base_params` train).  Returns (seconds, loss)."""
    import torch
    from oracle import llama_ref as R
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        cfgd = dict(cfg_kw)
        trains = lambda k: lora_scale is None or ".lora_" in k
        params = {k: (v.detach().to(dtype).clone().requires_grad_(True) if trains(k) else v.detach().to(dtype)) for k, v in base_params.items()}
        trainable = [v for k, v in params.items() if trains(k)]
        opt = torch.optim.Adam(trainable, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2) if with_optimizer else None
        if batch is not None:
            ids, mask, labels, pos = batch
        else:
            g = torch.Generator().manual_seed(2)
            ids = torch.randint(1000, 100000, (1, S), generator=g)
            mask = torch.ones(1, S)
            pos = torch.arange(S)[None]
            labels = torch.full((1, S), -100); labels[:, -20:] = ids[:, -20:]
        inv = R.llama3_inv_freq(cfgd["head_dim"], rope[0], rope[1])
        t0 = time.perf_counter()
        loss = R.llama_loss(params, cfgd, ids, mask, labels, pos, inv, lora_scale=lora_scale)
        loss.backward()
        if opt is not None:
            torch.nn.utils.clip_grad_norm_(trainable, 1.0)
            opt.step()
        return time.perf_counter() - t0, float(loss.detach().float())
    finally:
        torch.set_num_threads(old)


def _gate_object(what, hip_loss, ref_loss):
    return {"what": what, "hip_loss": hip_loss, "fp32_oracle_loss": ref_loss, "loss_rel_err_vs_fp32_oracle": abs(hip_loss - ref_loss) / abs(ref_loss), "tol": 1e-2}


def train_cpu_baseline(cfg_kw, S, threads=6, gate=None, lora_scale=None, rope=LLAMA3_ROPE, variants=True, label="16 layers"):
    """One train step of the same architecture in plain PyTorch on the host CPU (the reference's CPU path is HF transformers on ATen
    CPU kernels).  `value`: fp32, batch 1, torch.set_num_threads(6) -- the reference's own setting, ecg_byte/main.py:2 -- forward +
    backward.  `variants`: all host cores, with the clip + Adam step, and bf16 weights / activations.
    `gate` = (state_dict of the HIP model as fp32 host tensors, its batch's first sample, the HIP path's loss on that sample): the timed fp32 leg then runs on the
    HIP model's OWN weights (and adapters) and sample, and its loss is the full-size parity gate of SURVEY.md section 8d (HIP loss within 1e-2 rel of the fp32 restatement)."""
    import torch
    from oracle import llama_ref as R
    if gate is not None:
        base, batch, hip_loss = gate
    else:
        base, batch, hip_loss = R.random_params(dict(cfg_kw), seed=0, dtype=torch.float32), None, None      # generated once: 1.24 G normal variates take longer than a step
    dt, ref_loss = _cpu_train_once(cfg_kw, S, threads, torch.float32, False, base, batch, lora_scale, rope)
    n_all = max(1, min(os.cpu_count() or 1, 64))
    var = []
    if variants:
        for dtype, thr, opt in ((torch.float32, n_all, True), (torch.bfloat16, n_all, False), (torch.bfloat16, threads, False)):
            t, _ = _cpu_train_once(cfg_kw, S, thr, dtype, opt, base, batch, lora_scale, rope)
            var.append({"dtype": str(dtype).replace("torch.", ""), "threads": thr, "optimizer_step": opt, "samples_per_s": 1.0 / t})
    out = {"value": 1.0 / dt, "unit": "samples/s", "cores": threads, "kind": "port",
           "sample": f"1 sample (seq {S}) fwd+bwd{' (adapters only)' if lora_scale is not None else ''}, fp32, oracle/llama_ref.py, {dt:.1f} s"}
    if var:
        out["variants"] = {k: [v[k] for v in var] for k in var[0]}
    if hip_loss is not None:
        out["parity_gate"] = _gate_object(f"{label}: HIP model's weights, sample 0, training forward vs this fp32 leg", hip_loss, ref_loss)
    return out


def hip_gate_sample(model, opt, one, dev, lora_seed=None):
    """(host fp32 parameter dict in the oracle's names, the sample on the host, the HIP training-forward loss on it) for `train_cpu_baseline(gate=...)`.
    With LoRA on: dropout off for this one forward (the oracle has none) and lora_B drawn non-zero (peft's zero start would make the adapter branch a no-op
    and the gate blind to it); both are restored afterwards -- the timed steps start from peft's initialisation with dropout 0.05."""
    import torch
    sites = [s for layer in model.lora for s in layer.values()] if model.lora is not None else []
    saved_p = [s.p for s in sites]
    if sites:
        g = torch.Generator(device=dev).manual_seed(lora_seed or 5)
        with torch.no_grad():
            for s in sites:
                s.B.copy_((torch.randn(s.B.shape, device=dev, generator=g) * 0.02).to(torch.bfloat16) * s.bmask)
                s.p = 0.0
    opt.zero_grad()
    o = model(input_ids=one[0], attention_mask=one[1], labels=one[2], position_ids=one[3])
    hip_loss = float(o.loss.item())
    o.loss.backward()                                # (consumes the saved state; the gradients are dropped by the next zero_grad)
    opt.zero_grad()
    sd = {k: v.detach().float().cpu() for k, v in model._hf_named() if k != "lm_head.weight"}
    if sites:
        for name, t in model.lora_named():
            sd[name.replace("base_model.model.", "").replace(".default.weight", "")] = t.detach().float().cpu()
        with torch.no_grad():
            for s, p in zip(sites, saved_p):
                s.B.zero_()
                s.p = p
    return sd, tuple(t.cpu() for t in one), hip_loss


def hbm_kernel_report(dev, B, S, cfg, n_vocab, reps=10):
    """HBM GB/s of the memory-bound kernels of the train step at its shapes (RMSNorm, RoPE, SwiGLU, CE, Adam report
    GB/s against 8 TB/s instead of TFLOP/s).  Each kernel alone, `reps` launches between HIP events on the launch stream;
    bytes = what the kernel must read and write once (bf16 activations, fp32 statistics / moments)."""
    import torch
    from ecg_byte_amd import decoder_ops as ops
    T, H, I = B * S, cfg.hidden_size, cfg.intermediate_size
    Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    QKV = (Hq + 2 * Hkv) * D
    bf = lambda *sh: torch.randn(*sh, device=dev).to(torch.bfloat16)
    x, r, w = bf(T, H), bf(T, H), torch.ones(H, dtype=torch.bfloat16, device=dev)
    y, rstd, xs = ops.rmsnorm_fwd(x, w, 1e-5, residual=r)
    dwf = torch.zeros(H, dtype=torch.float32, device=dev)
    qkv = bf(T, QKV)
    pos = torch.arange(S, device=dev).repeat(B).float()
    fr = pos[:, None] * torch.rand(D // 2, device=dev)[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    gu, dh = bf(T, 2 * I), bf(T, I)
    rows = 4096
    logits = bf(rows, (n_vocab + 127) // 128 * 128)
    labels = torch.randint(0, n_vocab, (rows,), device=dev)
    inv, lsum = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    n_adam = 2 * I * H
    p_, g_ = bf(n_adam), bf(n_adam)
    m_, v_, acc = torch.zeros(n_adam, device=dev), torch.zeros(n_adam, device=dev), torch.ones(1, device=dev)
    cases = [
        ("rmsnorm_fwd+residual", lambda: ops.rmsnorm_fwd(x, w, 1e-5, residual=r), T * H * 2 * 4),          # read x, residual; write sum, y
        ("rmsnorm_bwd+residual", lambda: ops.rmsnorm_bwd(xs, w, rstd, y, dwf, dres=r), T * H * 2 * 4),   # read x, dy, dres; write dx
        ("rope", lambda: ops.rope_(qkv, cos, sin, Hq + Hkv, D, QKV), T * (Hq + Hkv) * D * 2 * 2 + T * D * 4),
        ("glu_fwd", lambda: ops.glu_fwd(gu), T * I * 2 * 3),
        ("glu_bwd", lambda: ops.glu_bwd(gu, dh), T * I * 2 * 5),
        ("ce_fwd_bwd", lambda: ops.ce_fwd_bwd_(logits, labels, inv, lsum, n_vocab), rows * logits.shape[1] * 2 * 2),
        ("adam_step", lambda: ops.adam_step_(p_, g_, m_, v_, acc, 1.0, 1e-4, 0.9, 0.99, 1e-8, 1e-2, 1), n_adam * (2 + 2 + 2 + 16)),
    ]
    out = []
    for name, fn, nbytes in cases:
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = nbytes / (ms * 1e-3) / 1e9
        out.append({"kernel": name, "ms": ms, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS})
    return out


def bench_batch_sweep(tk, pc, xd, L, sizes=(1, 64, 1024, 4096, 16384, 65536), reps=20):
    """the fused quantise + encode launch over B in {1, 64, 1024, 4096} records, and 16 384 / 65 536 (the bench batch of 4 096 is exactly ONE round
    of resident waves -- one record per wave -- so its launch ends with the slowest record of the slowest SIMD; a corpus pass runs many rounds per launch, and this is
    where that rate is measured rather than argued).  Inputs resident, HIP events on the launch stream; batches above the bench batch repeat its records."""
    import torch
    n = 12 * L
    out = []
    for B in sizes:
        free = torch.cuda.mem_get_info(xd.device)[0]
        if B > xd.shape[0] and B * n * 12 > 0.8 * free:            # 8 bytes of signal + 4 of worst-case ids per sample
            continue
        x = xd[:B] if B <= xd.shape[0] else xd.repeat((B + xd.shape[0] - 1) // xd.shape[0], 1, 1)[:B].contiguous()
        ids = torch.empty((B, n), dtype=torch.int32, device=xd.device)
        counts = torch.empty((B,), dtype=torch.int32, device=xd.device)
        r = reps if B <= 4096 else 5
        for _ in range(3):
            tk.quantize_encode(x, pc, ids_stride=n, out=(ids, counts))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(r):
            tk.quantize_encode(x, pc, ids_stride=n, out=(ids, counts))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / r
        toks = int(counts.sum().item())
        alg = B * 8 * n + 4 * toks
        out.append({"records": B, "ms": ms, "tokens_per_s": toks / (ms * 1e-3), "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        del x, ids, counts
        torch.cuda.empty_cache()
    return out


def gpu_serial():
    """The GPU's serial number by rocm-smi.  Called BEFORE this process touches the GPU: rocm-smi is a script (the child execs an interpreter), and on this pool a process
    that has initialised the GPU -- a forked child of one included, under `rocprofv3 --pmc` -- is refused such an exec."""
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "--showserial"], capture_output=True, text=True, timeout=20).stdout
        return next((ln.split(":")[-1].strip() for ln in txt.splitlines() if "Serial Number:" in ln), None)
    except Exception:
        return None


def device_selfcheck(serial):
    """Which GPU the line was measured on and whether it repeats a product bit for bit (torch.matmul, none of this repository's code: one GPU of the pool did not in
    round 4 -- EXPERIMENTS.md section R4, tests/test_gpu_00_selfcheck.py; the train legs' figures from such a device are not to be trusted)."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(2048, 512, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(3072, 512, device="cuda", generator=g) * 0.05).bfloat16()
    outs = [(x @ w.t()).clone() for _ in range(40)]
    differ = sum(not torch.equal(outs[0], o) for o in outs[1:])
    return {"serial": serial, "torch_matmul_launches_that_differ_from_the_first": differ, "of": 39, "repeats_bit_for_bit": differ == 0}


def bench_trainer(pc, L, x, num_merges=4000):
    """rust_bpe.byte_pair_encoding on the device for the C2 tokenizer's own corpus (2 000 synthetic records, seed 1).
    Algorithmic bytes per merge i: the ids read once and the survivors written once, two bytes an id (256 + 4 000 ids fit 16 bits): 2 N_i + 2 N_{i+1}; the
    N_i are recovered exactly from the result (undoing merge i splits every token 256 + i into its two children).  Rounds 3-4 priced the two-pass form's own
    traffic (12 bytes per id and merge: `frac_at_round4_bytes`, the yardstick of their 0.35)."""
    import torch
    from ecg_byte_amd.tokenizer import quantize
    from ecg_byte_amd.trainer import bpe_train_device
    n_records = x.shape[0]
    sym = quantize(torch.from_numpy(x).cuda(), pc).view(-1)
    text = (sym + 97).contiguous()
    del x, sym
    best = None
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, n_ids, pairs, n_done = bpe_train_device(text, num_merges)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    k, m = int(n_done.item()), int(n_ids.item())
    cnt = torch.bincount(ids[:m].long(), minlength=256 + k).cpu().numpy().astype(np.int64)
    pr = pairs[:k].cpu().numpy()
    N = [0] * (k + 1)
    N[k] = m
    for i in range(k - 1, -1, -1):
        c = int(cnt[256 + i])
        cnt[pr[i, 0]] += c
        cnt[pr[i, 1]] += c
        N[i] = N[i + 1] + c
    assert N[0] == text.numel(), (N[0], text.numel())
    alg = sum(2 * N[i] + 2 * N[i + 1] for i in range(k))
    alg_r4 = sum(8 * N[i] + 4 * N[i + 1] for i in range(k))
    return {"workload": f"{n_records} records of 12x{L} = {text.numel()} symbols, {num_merges} merges (tokenizer_c2.pkl's corpus)",
            "seconds": best, "merges_done": k, "final_ids": m,
            "algorithmic_bytes": alg, "GB/s": alg / best / 1e9, "frac_of_hbm_peak": alg / best / 1e9 / HBM_PEAK_GBS,
            "frac_at_round4_bytes": alg_r4 / best / 1e9 / HBM_PEAK_GBS}


def bench_preprocess(dev, n_records=4096):
    """the offline conditioning (filter chain, wavelet shrinkage, 500 -> 250 Hz, segments) of raw 12 x 5000 float64 records on
    the device.  Algorithmic bytes: 480 KB read + 240 KB written per record.  The stages are recursions along time, one sequence (record, lead) per lane:
    4 096 records are 768 waves -- one per SIMD; fewer records take the same time (1 024: the same 20 ms), so the batch is what an offline pass over a corpus
    of 200 000 records would use.  `stage_traffic` is what the stages move as they are written (every filtfilt is a forward and a backward sweep through HBM,
    the wavelet transform one, the spline two and a quarter), not the algorithmic bytes; `bound` says what the kernels actually run against."""
    import torch
    from ecg_byte_amd import preprocess_utils as pp, synth
    base = np.ascontiguousarray(synth.synth_ecg(64, 5000, seed=0).transpose(0, 2, 1))
    x = np.concatenate([base] * (n_records // 64)) + 0.01 * np.random.default_rng(0).standard_normal((n_records, 5000, 12))
    xd = torch.from_numpy(x).to(dev)

    def timed(xin, reps=3):
        o = pp.condition_records(xin, reorder=True, seg_len=1250)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            o = pp.condition_records(xin, reorder=True, seg_len=1250)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timed(xd)
    # is the fraction parallelism-limited?  4 096 records are 49 152 sequences = 768 waves for 1 024 SIMDs; 16 384 and 65 536 records fill them four and sixteen times over
    larger = {}
    for nr in (16384, 32768):
        if nr * 5000 * 12 * 8 * 8 > 0.7 * torch.cuda.mem_get_info(dev)[0]:       # (input, four planes of scratch, outputs)
            continue
        xl = xd.repeat(nr // n_records, 1, 1)
        larger[str(nr)] = timed(xl, reps=1) * n_records / nr
        del xl
        torch.cuda.empty_cache()
    alg = n_records * (5000 * 12 * 8 + 2500 * 12 * 8)
    samples = n_records * 5000 * 12
    # sweeps through HBM as the stages are written, 8 bytes read + 8 written per sample and sweep: 4 filters x (forward, backward); wavelet: ONE (round 3: a
    # workgroup per sequence keeps the bands in LDS -- the samples are read once and the result written once; the lane-per-sequence kernel moved ~5);
    # spline: tridiagonal forward sweep, then back substitution and evaluation in one (36 bytes per sample) = 2.25; every finite test is in the kernels
    sweeps = {"filter_chain": 8.0, "wavelet": 1.0, "resample": 2.25}
    traffic = sum(sweeps.values()) * 16 * samples
    ach = alg / (ms * 1e-3) / 1e9
    return {"workload": f"{n_records} raw records of 5000 x 12 float64: filter chain, wavelet shrinkage, resample, segments",
            "ms": ms, "records_per_s": n_records / (ms * 1e-3), "ms_per_4096_records_at": larger,
            # the roofline of the stage: ALGORITHMIC bytes (480 KB read + 240 KB written per record) over the measured time against the HBM peak
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": alg},
            # what the kernels actually move (NOT a roofline: every filtfilt pass is a sweep through HBM scratch because scipy's recursion is kept sample by sample,
            # bit for bit) -- `x_algorithmic` is the waste factor
            "implementation_traffic_x_algorithmic": traffic / alg,      # (8 filter sweeps + 1 wavelet + 2.25 resample of 16 bytes a sample: DESIGN.md section 9)
            }


def bench_c5(args, dev):
    """BASELINE config C5's shapes on one GPU: Gemma-2B dims (18 layers, hidden 2048, 8 query heads / 1 KV head of 256, MLP 16 384,
    vocab 256 000 + 256 + 3 500 + 3), seq 2048, batch 8, LoRA r16 (what the reference's script runs, ecg_byte/scripts/train_model.sh),
    then greedy generate of 128 tokens after a 600-token prompt (ecg_byte/models/llm.py:26-37).  Random init, synthetic ids."""
    import torch
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    B, S = 8, 2048
    V = 256000 + 256 + 3500 + 3
    cfg = DecoderConfig.gemma_2b(vocab_size=V, pad_token_id=V - 1)
    m = HipCausalLM(cfg, device=dev, seed=0)
    m.enable_lora(r=16, alpha=32, dropout=0.05)
    opt = m.make_optimizer()
    g = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(1000, 100000, (B, S), device=dev, generator=g)
    mask = torch.ones(B, S, device=dev); mask[:, :100] = 0; ids[:, :100] = cfg.pad_token_id
    pos = (torch.cumsum(mask, 1) - 1).clamp(min=0).long(); pos[mask == 0] = 0
    labels = torch.full((B, S), -100, device=dev); labels[:, -24:] = ids[:, -24:]

    def step():
        opt.zero_grad()
        out = m(input_ids=ids, attention_mask=mask, labels=labels, position_ids=pos)
        out.loss.backward()
        opt.step_and_update_lr()
        return out.loss

    gate = None
    if not args.no_cpu_baseline:      # full-depth parity gate: 18 layers, one row of the batch (S 2048), adapters non-zero, dropout off, before any update
        gate = hip_gate_sample(m, opt, [t[:1].contiguous() for t in (ids, mask, labels, pos)], dev, lora_seed=7)
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    e0.record()
    for _ in range(n):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    H, I, Lyr, D, Hq, Hkv = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    per_layer = H * (Hq + 2 * Hkv) * D + Hq * D * H + 3 * H * I
    tokens, head_rows = B * S, B * 24
    flops = 2 * (2 * Lyr * per_layer * tokens + 2 * H * V * head_rows) + 3 * Lyr * 2 * S * Hq * D * tokens   # frozen base: forward + input gradients
    out = {"workload": f"C5: Gemma-2B dims ({Lyr} layers, vocab {V}), seq {S}, batch {B}, LoRA r16, random init",
           "train": {"ms_per_step": ms, "samples_per_s": B / (ms * 1e-3), "tokens_per_s": tokens / (ms * 1e-3), "loss": float(loss.item()),
                     "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "flops_per_step": flops},
                     "max_memory_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}}
    if gate is not None:
        n_all = max(1, min(os.cpu_count() or 1, 64))
        cfg_kw = dict(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=Lyr, num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=D,
                      rms_norm_eps=cfg.rms_norm_eps, model_type="gemma")
        out["train"]["cpu_baseline"] = train_cpu_baseline(cfg_kw, S, threads=n_all, gate=gate, lora_scale=2.0, rope=(10000.0, None), variants=False,
                                                          label="18 layers, LoRA (B != 0, no dropout)")
        del gate
    m.eval()
    prompt = ids[:1, -600:].contiguous()
    pm = torch.ones_like(prompt, dtype=torch.float32)

    def gen(n_new, merge=False):
        best = None
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.generate(input_ids=prompt, attention_mask=pm, max_new_tokens=n_new, pad_token_id=V - 1, merge_adapters=merge)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best
    t_all, t_few = gen(128), gen(8)                      # (8, not 1: the replayed-graph path needs more than two new tokens)
    per_token = (t_all - t_few) / 120.0                  # one decode step
    prefill = t_few - 8 * per_token                      # 600-token prompt through the batch kernels + the first token
    # what one decode step must read: every projection weight, the tied head over the padded vocabulary, the adapters, and the KV cache rows so far (mean over the 128 steps)
    v_pad = (V + 127) // 128 * 128
    w_bytes = 2 * (Lyr * per_layer + v_pad * H) + 2 * Lyr * 16 * (H + (Hq + 2 * Hkv) * D + Hq * D + H + H + 2 * I + I + H)
    kv_bytes = Lyr * 2 * Hkv * D * 2 * (600 + 64)
    out["generate"] = {"prompt_tokens": 600, "new_tokens": 128, "batch": 1, "seconds": t_all, "tokens_per_s": 128 / t_all,
                       "prefill_s": prefill, "decode_ms_per_token": per_token * 1e3, "decode_tokens_per_s": 1.0 / per_token,
                       "roofline": {"bound": "hbm", "bytes_per_token": w_bytes + kv_bytes, "achieved": (w_bytes + kv_bytes) / per_token / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": (w_bytes + kv_bytes) / per_token / 1e9 / HBM_PEAK_GBS}}
    # opt-in merged-adapter mode (peft's merge_and_unload for inference: W + (alpha / r) B A folded once per call, not bit-identical to the unmerged step)
    tm_all, tm_few = gen(128, True), gen(8, True)
    per_m = (tm_all - tm_few) / 120.0
    wm_bytes = 2 * (Lyr * per_layer + v_pad * H)
    out["generate"]["merged_adapters"] = {"seconds": tm_all, "tokens_per_s": 128 / tm_all, "decode_ms_per_token": per_m * 1e3, "decode_tokens_per_s": 1.0 / per_m,
                                          "roofline_frac": (wm_bytes + kv_bytes) / per_m / 1e9 / HBM_PEAK_GBS}
    del m, opt
    torch.cuda.empty_cache()
    return out


def bench_c1(args, dev, cpu=True):
    """BASELINE config C1 beside the headline workload: PTB-XL 100 Hz records (12 x 1000), the 1 000-merge tokenizer, GPT-2-small,
    batch 4.  HIP leg: quantise + encode + assemble on the device, then HipGPT2LM forward (loss only, eval mode) -- the reference's C1
    is a FORWARD pass; CPU leg: the rust_bpe port (oracle/ecgb_oracle.c, trie rebuilt per record as the reference does) and the
    PyTorch CPU forward of the same architecture (oracle/gpt2_ref.py) with the reference's 6 threads, with all host cores, and in bf16."""
    import torch
    from helpers import load_tokenizer
    from ecg_byte_amd import synth
    from ecg_byte_amd.data_loader import BatchAssembler
    from ecg_byte_amd.gpt2 import GPT2Config, HipGPT2LM
    from ecg_byte_amd.tokenizer import HipTokenizer
    vocab, merges, pc = load_tokenizer("c1")
    B, L, S = 4, 1000, 1024
    keys = list(vocab.keys())
    base = 50257
    n_vocab = base + len(keys) + 3
    lut = np.zeros(max(keys) + 1, dtype=np.int32)
    lut[keys] = base + np.arange(len(keys))
    sig_start, sig_end, pad, bos, eos = n_vocab - 3, n_vocab - 2, n_vocab - 1, 50256, 50256
    tk = HipTokenizer(merges)
    asm = BatchAssembler(tk, lut, pad, bos, eos, sig_start, sig_end, S - 4, device=dev)
    model = HipGPT2LM(GPT2Config(vocab_size=n_vocab, pad_token_id=pad), device=dev, seed=0).eval()
    x_host = synth.synth_ecg(B, L, seed=0, start=20_000_000)
    x = torch.from_numpy(x_host).to(dev)
    rng = np.random.default_rng(3)
    qs = [rng.integers(1000, 50000, size=int(rng.integers(8, 25))).tolist() for _ in range(B)]
    ans = [rng.integers(1000, 50000, size=int(rng.integers(4, 33))).tolist() for _ in range(B)]

    def step():
        batch = asm(x, pc, qs, ans)
        with torch.no_grad():
            return model(input_ids=batch["tokenized_signal"], attention_mask=batch["attn_mask"],
                         labels=batch["quantized_signal_ids_input"], position_ids=batch["position_ids"]).loss, batch

    for _ in range(3):
        loss, batch = step()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    reps = 20
    e0.record()
    for _ in range(reps):
        asm(x, pc, qs, ans)
    e1.record()
    for _ in range(reps):
        loss, batch = step()
    e2.record()
    torch.cuda.synchronize()
    enc_ms, step_ms = e0.elapsed_time(e1) / reps, e1.elapsed_time(e2) / reps
    out = {"workload": f"C1: 12x{L}, {len(merges)} merges, GPT-2-small (vocab {n_vocab}), seq {S}, batch {B}, forward",
           "hip": {"quantise_encode_assemble_ms": enc_ms, "encode_plus_forward_ms": step_ms, "samples_per_s": B / (step_ms * 1e-3),
                   "loss": float(loss.item())}}
    if cpu:
        from oracle import gpt2_ref as G
        from oracle import oracle as O
        p1, p99 = pc["percentile_1"], pc["percentile_99"]
        t0 = time.perf_counter(); n_enc = 0
        while time.perf_counter() - t0 < 2.0:
            for b in range(B):
                O.encode_text(O.symbols_to_text(O.quantize(x_host[b], p1, p99)), merges)
            n_enc += 1
        enc_cpu = (time.perf_counter() - t0) / n_enc
        cfgd = dict(vocab_size=n_vocab, n_positions=1024, n_embd=768, n_layer=12, n_head=12)
        ids, mask = batch["tokenized_signal"].cpu(), batch["attn_mask"].cpu()
        labels, pos = batch["quantized_signal_ids_input"].cpu(), batch["position_ids"].cpu()
        variants = []
        n_all = max(1, min(os.cpu_count() or 1, 64))
        old = torch.get_num_threads()
        try:
            base = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k != "lm_head.weight"}     # the HIP model's own weights: its loss is the gate's
            ref_loss = None
            for dtype, thr in ((torch.float32, 6), (torch.float32, n_all), (torch.bfloat16, n_all)):
                torch.set_num_threads(thr)
                params = {k: v.to(dtype) for k, v in base.items()}
                with torch.no_grad():
                    l0 = float(G.gpt2_loss(params, cfgd, ids, mask, labels, pos).float())    # warm-up
                    ref_loss = l0 if ref_loss is None else ref_loss                           # (the first variant: fp32)
                    t0 = time.perf_counter(); k = 0
                    while time.perf_counter() - t0 < 4.0:
                        G.gpt2_loss(params, cfgd, ids, mask, labels, pos)
                        k += 1
                    fwd = (time.perf_counter() - t0) / k
                variants.append({"dtype": str(dtype).replace("torch.", ""), "threads": thr, "forward_s": fwd,
                                 "samples_per_s": B / (enc_cpu + fwd)})
        finally:
            torch.set_num_threads(old)
        out["cpu_baseline"] = {"value": variants[0]["samples_per_s"], "unit": "samples/s", "cores": 6, "kind": "port",
                               "encode_s_per_batch_1_core": enc_cpu,
                               "sample": f"batch of {B}: rust_bpe port (1 core) + GPT-2-small forward, torch CPU fp32, 6 threads",
                               "variants": {k: [v[k] for v in variants] for k in variants[0]}}
        out["parity_gate"] = _gate_object("12 layers: HIP model's weights, the timed batch, forward vs the fp32 CPU leg",
                                          out["hip"]["loss"], ref_loss)
        out["speedup_vs_cpu_6_threads"] = out["hip"]["samples_per_s"] / out["cpu_baseline"]["value"]
    del model
    torch.cuda.empty_cache()
    return out


LOADER_WORDS = ("sinus rhythm normal ecg atrial fibrillation with rapid ventricular response bradycardia otherwise left bundle branch block premature complexes "
                "axis deviation possible inferior infarct age undetermined nonspecific t wave abnormality prolonged qt borderline first degree av").split()
LOADER_QUESTION = "Could you please help me explain my ECG?"


def write_loader_files(x, pc, vocab, merges, n):
    """The reference's on-disk training set (preprocess_utils.py:215-226, file_utils.py:30-48): `ecg/ecg_{i}_{j}.npy` (12 x L float64) + `text/text_{i}_{j}.json`
    pairs, the percentiles .npy and the tokenizer pickle, under a fresh directory in /tmp.  Returns the directory."""
    import pickle
    import tempfile
    from ecg_byte_amd import file_utils as F
    root = tempfile.mkdtemp(prefix="ecgb_loader_", dir="/tmp")
    os.makedirs(os.path.join(root, "ecg")); os.makedirs(os.path.join(root, "text"))
    rng = np.random.default_rng(11)
    for i in range(min(n, x.shape[0])):
        np.save(os.path.join(root, "ecg", f"ecg_{i}_0.npy"), x[i])
        words = rng.choice(LOADER_WORDS, size=int(rng.integers(4, 33)))
        with open(os.path.join(root, "text", f"text_{i}_0.json"), "w") as f:
            json.dump(" ".join(words), f)
    F.save_percentiles(os.path.join(root, "percentiles.npy"), pc["percentile_1"], pc["percentile_99"])
    with open(os.path.join(root, "tok.pkl"), "wb") as f:
        pickle.dump((vocab, merges), f)
    return root


def bench_loader(model, opt, dev, root, B, S, steps, resident_ms):
    """The path the reference's loop actually runs (ecg_byte/data_loader.py:52-83, main.py:249-259: np.load + json + tokenizer inside __getitem__, num_workers=0), here
    through `DeviceBatchLoader` (reader threads, pinned staging, copy on a side stream, ONE quantise + encode + assemble per batch on the device): the same LoRA train
    step fed from the files `write_loader_files` wrote, against the figure with the batch resident in HBM.  `loader_stall_ms_per_step` = host time the training thread
    spent waiting for a batch the readers had not finished.  The files were just written: they are read from the page cache, not from a disk."""
    import torch
    from types import SimpleNamespace
    from helpers import WordTokenizer
    from ecg_byte_amd import file_utils as F
    from ecg_byte_amd.data_loader import DeviceBatchLoader, ECGTokenDataset
    vocab, merges = F.load_vocab_and_merges(os.path.join(root, "tok.pkl"))
    tok = WordTokenizer(LOADER_QUESTION.split() + list(LOADER_WORDS))
    tok.add_tokens([f"signal_{k}" for k in vocab.keys()])                  # main.py:144-150
    tok.add_tokens(["<sig_start>"], special_tokens=True)
    tok.add_tokens(["<sig_end>"], special_tokens=True)
    tok.add_special_tokens({"pad_token": "<pad>"})
    sig, txt = F.align_signal_text_files(os.path.join(root, "ecg"), os.path.join(root, "text"))
    ns = SimpleNamespace(percentiles=os.path.join(root, "percentiles.npy"), dataset="ptb_500", inference=False, pad_to_max=S - 4, dis=False, dev=False,
                         toy=True, device=str(dev))
    ds = ECGTokenDataset(sig, txt, vocab, merges, tokenizer=tok, args=ns)
    runs = []
    for workers in (1, 4):
        ld = DeviceBatchLoader(ds, batch_size=B, shuffle=True, seed=0, workers=workers, drop_last=True)
        it = iter(ld)

        def run(n):
            for _ in range(n):
                b = next(it)
                opt.zero_grad()
                o = model(input_ids=b["tokenized_signal"], attention_mask=b["attn_mask"], labels=b["quantized_signal_ids_input"], position_ids=b["position_ids"])
                o.loss.backward()
                opt.step_and_update_lr()
        run(2)
        torch.cuda.synchronize()
        stall0, t0 = ld.stall_s, time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        runs.append({"reader_threads": workers, "ms_per_step_from_disk": ms, "loader_stall_ms_per_step": (ld.stall_s - stall0) / steps * 1e3})
        it.close()
    best = min(runs, key=lambda r: r["ms_per_step_from_disk"])
    sys.stderr.write("bench.py: loader runs " + json.dumps(runs) + "\n")
    return {"workload": f"{len(sig)} .npy + .json pairs in /tmp (page cache) -> DeviceBatchLoader (batch {B}, shuffle) -> the C3 LoRA step",
            "ms_per_step_resident": resident_ms, "ms_per_step_from_disk": best["ms_per_step_from_disk"], "loader_stall_ms_per_step": best["loader_stall_ms_per_step"],
            "reader_threads": best["reader_threads"], "steps": steps}


def bench_train(args, tk, vocab, merges, pc, world, rank, dev, x_host, loader_dir=None):
    """Train samples/s of Llama-3.2-1B (seq 1024, bf16, B per GPU) on the HIP decoder, batches built
    through the real front end: synthetic ECG -> quantise+encode -> LUT -> assemble."""
    import torch
    import torch.distributed as dist
    from ecg_byte_amd.data_loader import BatchAssembler
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    from ecg_byte_amd.parallel import GradAllReduce
    B, S, L = args.train_batch, 1024, args.L
    base = 128256
    keys = list(vocab.keys())
    n_vocab = base + len(keys) + 3                       # main.py:144-151: signal tokens, <sig_start>, <sig_end>, <pad>
    lut = np.zeros(max(keys) + 1, dtype=np.int32)
    lut[keys] = base + np.arange(len(keys))               # ids follow the pickled dict order
    sig_start, sig_end, pad = n_vocab - 3, n_vocab - 2, n_vocab - 1
    bos, eos = 128000, 128001
    cfg = DecoderConfig.llama_3_2_1b(vocab_size=n_vocab, pad_token_id=pad)
    model = HipCausalLM(cfg, device=dev, seed=0)
    model.full_logits = bool(args.full_logits)
    if args.lora:
        model.enable_lora(r=16, alpha=32, dropout=0.05)      # ecg_byte/main.py:131-138
    if dist.is_initialized():
        model.grad_sync = GradAllReduce(single_rank_collectives=True, time_exposed=True)
    # Adam(0.9, 0.99, 1e-8, wd 1e-2) + Noam(500) + clip 1.0.  --overlap-optimizer: the parameter updates on a side stream under the next step's forward (same
    # bits: tests/test_gpu_fullshape.py::test_overlapped_optimizer_step_is_the_plain_step_bit_for_bit) -- measured 189.2 -> 188.4 ms, not the 5.7 ms of Adam
    # traffic: the GEMMs it hides under are clocked by the power budget, which the extra HBM traffic shares; off by default
    opt = model.make_optimizer(overlap=args.overlap_optimizer)
    asm = BatchAssembler(tk, lut, pad, bos, eos, sig_start, sig_end, S - 4, device=dev)
    rng = np.random.default_rng(2 + rank)
    x = torch.from_numpy(x_host).to(dev)
    qs = [rng.integers(1000, 100000, size=int(rng.integers(8, 25))).tolist() for _ in range(B)]
    ans = [rng.integers(1000, 100000, size=int(rng.integers(4, 33))).tolist() for _ in range(B)]

    def step():
        batch = asm(x, pc, qs, ans)                       # quantise + encode + assemble on the device
        opt.zero_grad()
        out = model(input_ids=batch["tokenized_signal"], attention_mask=batch["attn_mask"],
                    labels=batch["quantized_signal_ids_input"], position_ids=batch["position_ids"])
        out.loss.backward()
        opt.step_and_update_lr()
        return out.loss

    gate = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # full-size parity gate: before any update, the training-mode forward of the 16-layer model on the first sample of its own batch;
        # the fp32 restatement runs on the same weights (and adapters) and sample inside train_cpu_baseline (its timed fp32 leg)
        b0 = asm(x, pc, qs, ans)
        one = [b0[k][:1].contiguous() for k in ("tokenized_signal", "attn_mask", "quantized_signal_ids_input", "position_ids")]
        gate = hip_gate_sample(model, opt, one, dev)
        del b0, one
    for _ in range(max(1, min(args.warmup, 5))):   # a train step is 300x an encode step: a few warm-up steps are enough
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.train_steps):
        loss = step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    n_valid = sum(len(a) + 1 for a in ans)                # answer tokens + eos carry labels
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    sec = wall / args.train_steps
    comm = None
    if dist.is_initialized() and model.grad_sync is not None:
        gs = model.grad_sync
        comm = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "collectives_per_step": gs.collectives / max(1, gs.steps),
                "exposed_ms_per_step": gs.exposed_ms() / max(1, gs.steps),
                "note": "exposed = time the compute stream waits in GradAllReduce.finish() for buckets still in flight (HIP events around the waits)"}
    # algorithmic FLOPs: 3 x (2 x matmul params x tokens + causal attention), loss head over the rows it runs on
    H, I, Lyr = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    per_layer = H * (cfg.num_attention_heads * cfg.head_dim + 2 * cfg.num_key_value_heads * cfg.head_dim) + cfg.num_attention_heads * cfg.head_dim * H + 3 * H * I
    tokens = B * S
    head_rows = tokens if model.full_logits else n_valid
    flops = 3 * (2 * Lyr * per_layer * tokens + Lyr * 2 * S * H * tokens + 2 * H * n_vocab * head_rows)
    if args.lora:   # frozen base: no weight-gradient products (2/3 of the matmul work remains), adapters add ~1 %
        flops = 2 * (2 * Lyr * per_layer * tokens + 2 * H * n_vocab * head_rows) + 3 * Lyr * 2 * S * H * tokens
    achieved = flops / sec / 1e12
    out = {"metric": "train_samples_per_sec", "value": B * world / sec, "unit": "samples/s", "ms_per_step": sec * 1e3,
           "steps": args.train_steps, "dtype": "bf16", "final_loss": float(loss.item()),
           "config": {"workload": f"C3: Llama-3.2-1B dims (vocab {n_vocab}), seq {S}, batch {B}/GPU, "
                                  f"{'LoRA r16' if args.lora else 'full fine-tune'}, random init; batches built on device",
                      "loss_head_rows": "all" if model.full_logits else "labelled",
                      "parallelism": f"dp{world}" + (f" (bucketed async all-reduce of the flat gradient buffer, backend {dist.get_backend()})" if dist.is_initialized() else "")},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "gemm_nt_w4_kernel (bf16 MFMA 16x16x32) + attention",
                        "flops_per_step": flops, "step_ms_hip_events": ev0.elapsed_time(ev1) / args.train_steps}}
    if comm is not None:
        out["gradient_exchange"] = comm
    if rank == 0 and world == 1 and loader_dir is not None:
        out["loader"] = bench_loader(model, opt, dev, loader_dir, B, S, args.train_steps, sec * 1e3)
    if rank == 0 and not args.lora and not getattr(args, "no_hbm_report", False):
        del model, opt
        model = opt = None
        torch.cuda.empty_cache()
        rep = hbm_kernel_report(dev, B, S, cfg, n_vocab)      # columns (the line has 8 KB): kernel, ms per launch, fraction of the 8 TB/s spec (GB/s = frac x 8000)
        out["hbm_bound_kernels"] = {k: [r[k] for r in rep] for k in ("kernel", "ms", "frac")}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baselines are an N = 1 item
        cfg_kw = dict(vocab_size=n_vocab, hidden_size=H, intermediate_size=I, num_hidden_layers=Lyr,
                      num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                      head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_norm_eps)
        out["cpu_baseline"] = train_cpu_baseline(cfg_kw, S, gate=gate, lora_scale=2.0 if args.lora else None, variants=not args.lora,
                                                 label="16 layers, LoRA (B != 0, no dropout)" if args.lora else "16 layers")
    del model, opt
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="records per GPU")
    ap.add_argument("--L", type=int, default=5000, help="samples per lead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the train-step part (encode metric only)")
    ap.add_argument("--train-batch", type=int, default=32, help="samples per GPU per train step")
    ap.add_argument("--train-steps", type=int, default=10, help="timed steps of every train leg (the boxes differ by 1-2 %: ten steps per figure, not two or three)")
    ap.add_argument("--full-logits", action="store_true", help="loss head over every row, as the reference materialises it")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 object (Gemma-2B dims LoRA step + generate)")
    ap.add_argument("--no-c1", action="store_true", help="skip the C1 object (12x1000 records + GPT-2-small forward, batch 4)")
    ap.add_argument("--no-lora-leg", action="store_true", help="full fine-tune leg only (profiling: one mode per kernel trace)")
    ap.add_argument("--overlap-optimizer", action="store_true", help="train legs: parameter updates on a side stream under the next forward (A/B; measured +0.4 %)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects of the line (batch sweep, tokenizer trainer, offline conditioning, full-logits leg)")
    ap.add_argument("--lora", action="store_true", help="train LoRA adapters (r16, alpha 32, dropout 0.05; frozen base) as the reference's script does")
    args = ap.parse_args()
    # --gpus N must be what the launcher started: checked before anything is generated or initialised (exit 2, no line); the process group's own size is checked again below
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start it as python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}\n")
        sys.exit(2)

    # ONE line on stdout: libraries print there too (RCCL's version banner at the first collective under torch.distributed.run lands on file descriptor 1, in front of
    # the line).  Everything this process and its children write to stdout goes to stderr from here on; the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from helpers import load_tokenizer
    from ecg_byte_amd.tokenizer import HipTokenizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # host-side synthetic data first: the generator forks worker processes, which must happen before
    # this process initialises HIP / RCCL
    workers = int(os.environ.get("ECGB_BENCH_WORKERS", max(1, min(8, (os.cpu_count() or 1) // max(1, world)))))   # 1: no fork pool (profiler runs)
    x = make_signals(args.batch, args.L, seed=0, start=rank * args.batch, workers=workers)   # rank r owns records rB..rB+B-1
    loader_dir = None
    serial = gpu_serial() if rank == 0 else None               # (before HIP is up in this process)
    x_train = None if args.no_train else make_signals(args.train_batch, args.L, seed=0,
                                                      start=10_000_000 + rank * args.train_batch, workers=workers)
    # the tokenizer trainer's corpus (2 000 records, seed 1: what tests/golden/tokenizer_c2.pkl was trained on) -- generated here, before HIP is up, like the rest
    x_corpus = make_signals(2000, args.L, seed=1, start=0, workers=workers) if (rank == 0 and world == 1 and not args.no_extras and args.L == 5000) else None
    # Test hooks (tests/test_gpu_pipeline.py runs the N = 2 path on a one-GPU box): ECGB_BENCH_BACKEND=gloo replaces RCCL,
    # ECGB_BENCH_ONE_DEVICE=1 puts every rank on cuda:0.  Neither is set in a measured run.
    backend = os.environ.get("ECGB_BENCH_BACKEND", "nccl")
    if os.environ.get("ECGB_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    # started by torch.distributed.run (RANK in the environment): the process group is initialised even for ONE rank, and the
    # train legs then send their gradient buckets through RCCL all the same -- the one-GPU check of the N > 1 code path
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or launched:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    # A scaling run must be what it says: exit non-zero (no JSON line) if the process group is not --gpus ranks wide, or if two RCCL ranks sit on one device
    # (the gloo one-device test hook is exempt: it exists to run the N = 2 code path on a one-GPU box and is never a measured run).
    pg_world = dist.get_world_size() if dist.is_initialized() else 1
    if world != args.gpus or pg_world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}, process group of {pg_world} ranks\n")
        sys.exit(2)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    pg_backend = dist.get_backend() if dist.is_initialized() else None
    if dist.is_initialized() and world > 1 and pg_backend == "nccl" and os.environ.get("ECGB_BENCH_ONE_DEVICE") != "1":
        pr = torch.cuda.get_device_properties(dev)
        mine = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', -1):02x}/{getattr(pr, 'uuid', local_rank)}"
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        if len(set(seen)) != world:
            sys.stderr.write(f"bench.py: {world} RCCL ranks on {len(set(seen))} distinct devices ({seen}): not a {world}-GPU run\n")
            sys.exit(3)

    tag = "c2" if args.L == 5000 else "c1"
    vocab, merges, pc = load_tokenizer(tag)
    if rank == 0 and world == 1 and not args.no_extras and not args.no_train and not args.no_lora_leg and not args.lora and args.L == 5000:
        loader_dir = write_loader_files(x, pc, vocab, merges, 2048)       # the disk -> step leg's training set (the encode batch's first records)
    tk = HipTokenizer(merges)
    B, L = args.batch, args.L
    n = 12 * L
    xd = torch.from_numpy(x).to(dev)
    del x
    ids = torch.empty((B, n), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)

    def step():
        tk.quantize_encode(xd, pc, ids_stride=n, out=(ids, counts))

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    # HIP events on the launch stream (torch's current stream is the one the C ABI is given)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps
    tokens_rank = int(counts.sum().item())

    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        tt = torch.tensor([tokens_rank], dtype=torch.int64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        tokens_total = int(tt.item())
    else:
        tokens_total = tokens_rank

    sweep = None
    if rank == 0 and world == 1 and not args.no_extras and L == 5000:
        sweep = bench_batch_sweep(tk, pc, xd, L)
    c1 = None
    if rank == 0 and world == 1 and not args.no_c1:
        c1 = bench_c1(args, dev, cpu=not args.no_cpu_baseline)
    train = None
    if not args.no_train:
        del xd, ids
        torch.cuda.empty_cache()
        train = bench_train(args, tk, vocab, merges, pc, world, rank, dev, x_train)
        if not args.lora and not args.no_lora_leg:   # SURVEY.md §8d asks for both: full fine-tune (BASELINE C3 wording) and LoRA r16 (what the reference's script runs)
            import copy
            largs = copy.copy(args)
            largs.lora, largs.train_steps = True, args.train_steps
            lora = bench_train(largs, tk, vocab, merges, pc, world, rank, dev, x_train, loader_dir=loader_dir)
            train["lora_r16"] = {k: lora[k] for k in ("value", "unit", "ms_per_step", "steps", "final_loss", "roofline", "cpu_baseline", "loader") if k in lora}
            train["lora_r16"]["roofline"].pop("kernel", None)
            train["lora_r16"]["workload"] = "same step, LoRA r16 on q,k,v,o,gate,up,down (main.py:131-138)"

    if train is not None and not args.lora and not args.full_logits and not args.no_extras:
        # the reference-equivalent loss head (modeling_llama.py:1209-1213 materialises every row's logits): the same step with the head over all B x S rows
        import copy
        fargs = copy.copy(args)
        fargs.full_logits, fargs.no_cpu_baseline, fargs.train_steps, fargs.no_hbm_report = True, True, args.train_steps, True
        full = bench_train(fargs, tk, vocab, merges, pc, world, rank, dev, x_train)
        train["full_logits"] = {k: full[k] for k in ("value", "unit", "ms_per_step", "steps", "final_loss", "roofline")}
        train["full_logits"]["roofline"].pop("kernel", None)
        train["full_logits"]["workload"] = "same full fine-tune step, loss head over all rows (as the reference)"
    if train is not None:
        # per-kernel MFMA-pipe busy fractions come from counter passes of their own (committed, not measured by this run): the line names the files only
        f = _latest_profile("train_pmc.json")
        if f:
            train["roofline"]["mfma_busy_source"] = os.path.relpath(f, ROOT)
        f = _latest_profile("train_pmc_lora.json")
        if f and "lora_r16" in train:
            train["lora_r16"]["roofline"]["mfma_busy_source"] = os.path.relpath(f, ROOT)
    c5 = None
    if rank == 0 and world == 1 and not args.no_c5 and not args.no_train:
        c5 = bench_c5(args, dev)
    extras = {}
    if rank == 0:
        extras["device"] = device_selfcheck(serial)
    if rank == 0 and world == 1 and not args.no_extras and L == 5000:
        extras["trainer"] = bench_trainer(pc, L, x_corpus)
        del x_corpus
        extras["preprocess"] = bench_preprocess(dev)
    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        records_total = B * world
        alg_bytes = B * (8 * n) + 4 * tokens_rank          # per launch on one GPU
        achieved = alg_bytes / (dev_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic_bytes() if (B == 4096 and L == 5000) else (None, None)
        out = {
            "metric": "ecg_tokens_per_sec_encode", "value": tokens_total / (wall / args.steps),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64->u8->u32", "data": "synthetic",
            "rccl_ranks": pg_world if pg_backend == "nccl" else 0, "backend": pg_backend,      # the process group this line was measured under (None: one plain process)
            "config": {"workload": f"C2: PTB-XL-shaped 12x{L} float64 records, vocab {len(merges)} merges, "
                                   f"quantise+encode, {B} records/GPU", "records_per_gpu": B,
                       "samples_per_record": n, "merges": len(merges), "parallelism": f"dp{world} (sharded records, no collective)"},
            "symbols_per_s": records_total * n / (wall / args.steps),
            "records_per_s": records_total / (wall / args.steps),
            "tokens_per_record": tokens_total / records_total,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "encode_flow_kernel<INPUT_F64> (fused quantise+encode, one launch per step)",
                         "kernel_ms": dev_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(merges, pc, L, seed=0)
        if sweep is not None:      # columns: records per launch, ms per launch, token ids/s, fraction of the HBM roofline (algorithmic bytes)
            out["batch_sweep"] = {k: [r[k] for r in sweep] for k in ("records", "ms", "frac")}
        out.update(extras)
        if c1 is not None:
            out["c1"] = c1
        if c5 is not None:
            out["c5"] = c5
        if train is not None:
            out["train"] = train
        out = with_top_level_scalars(out)
        line = json.dumps(compact(out), separators=(",", ":"))
        if len(line) > 8000:                                    # the driver keeps the top level and an 8 KB tail: shed the bulkiest detail objects, never the scalars
            for path in (("c1", "cpu_baseline", "variants"), ("train", "cpu_baseline", "variants"),
                         ("cpu_baseline", "sample"), ("batch_sweep",), ("train", "hbm_bound_kernels")):
                d = out
                for k in path[:-1]:
                    d = d.get(k, {}) if isinstance(d, dict) else {}
                if isinstance(d, dict) and path[-1] in d:
                    sys.stderr.write("bench.py: line over 8 KB, moved to stderr: " + ".".join(path) + " = " + json.dumps(compact(d[path[-1]])) + "\n")
                    d[path[-1]] = "see stderr (line kept under 8 KB)"
                    line = json.dumps(compact(out), separators=(",", ":"))
                    if len(line) <= 8000:
                        break
        real_stdout.write(line + "\n")
        real_stdout.flush()
    if loader_dir is not None:
        import shutil
        shutil.rmtree(loader_dir, ignore_errors=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
