"""Dev-only: the C2 corpus (2 000 synthetic records, seed 1; 4 000 merges) through ecgb_bpe_train_hip with the next merge's row maxima inside the merge's launch
(round 6, one launch per merge) and as a launch of their own (round 5, the default), best of three each, and that both give the same merges.
Usage: python scripts/dev_trainer_ab.py [num_merges]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from ecg_byte_amd import trainer
from ecg_byte_amd.tokenizer import quantize

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
x = bench.make_signals(2000, 5000, seed=1, start=0, workers=8)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_tokenizer
_, _, pc = load_tokenizer("c2")
text = (quantize(torch.from_numpy(x).cuda(), pc).view(-1) + 97).contiguous()
res = {}
for fused in (1, 0, 1, 0):
    trainer.set_train_fused(bool(fused))
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, n_ids, pairs, n_done = trainer.bpe_train_device(text, k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = (ids[: int(n_ids)].cpu(), pairs[: int(n_done)].cpu())
    if res:
        assert torch.equal(out[0], res["ids"]) and torch.equal(out[1], res["pairs"]), "forms disagree"
    res = {"ids": out[0], "pairs": out[1]}
    print(f"fused={fused}: {best * 1e3:.2f} ms for {int(n_done)} merges ({best / int(n_done) * 1e6:.2f} us a merge), {int(n_ids)} ids left", flush=True)
trainer.set_train_fused(False)
