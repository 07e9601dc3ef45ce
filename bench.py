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


def measured_traffic_bytes():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r*/hbm_pmc.json; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for
    16 B/lane coalesced reads).  None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_pmc.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    return (2.0 * d["FETCH_SIZE"]["per_launch_mean"] + d["WRITE_SIZE"]["per_launch_mean"]) * 1024.0


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
    return {
        "value": toks_f / dt_f, "unit": "tokens/s", "cores": 1, "kind": "port",
        "sample": f"{n_f} records of 12x{L} (seed {seed}) in {dt_f:.1f} s, trie rebuilt per call as the "
                  f"reference does; oracle/ecgb_oracle.c",
        "records_per_s": n_f / dt_f,
        "value_trie_built_once": toks / dt_once,
        "records_per_s_trie_built_once": n_once / dt_once,
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="records per GPU")
    ap.add_argument("--L", type=int, default=5000, help="samples per lead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from helpers import load_tokenizer
    from ecg_byte_amd.tokenizer import HipTokenizer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    tag = "c2" if args.L == 5000 else "c1"
    _, merges, pc = load_tokenizer(tag)
    tk = HipTokenizer(merges)
    B, L = args.batch, args.L
    n = 12 * L
    workers = max(1, min(8, (os.cpu_count() or 1) // max(1, world)))
    x = make_signals(B, L, seed=0, start=rank * B, workers=workers)   # rank r owns records rB..rB+B-1
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

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        records_total = B * world
        alg_bytes = B * (8 * n) + 4 * tokens_rank          # per launch on one GPU
        achieved = alg_bytes / (dev_ms * 1e-3) / 1e9
        out = {
            "metric": "ecg_tokens_per_sec_encode", "value": tokens_total / (wall / args.steps),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64->u8->u32", "data": "synthetic",
            "config": {"workload": f"C2: PTB-XL-shaped 12x{L} float64 records, vocab {len(merges)} merges, "
                                   f"quantise+encode, {B} records/GPU", "records_per_gpu": B,
                       "samples_per_record": n, "merges": len(merges), "parallelism": f"dp{world} (sharded records, no collective)"},
            "symbols_per_s": records_total * n / (wall / args.steps),
            "records_per_s": records_total / (wall / args.steps),
            "tokens_per_record": tokens_total / records_total,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic_bytes() if (B == 4096 and L == 5000) else None,
                         "kernel": "encode_wave_kernel<64, INPUT_F64> (fused quantise+encode, one launch per step)",
                         "kernel_ms": dev_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(merges, pc, L, seed=0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
