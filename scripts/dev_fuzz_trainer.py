"""Dev-only: randomised parity soak of the HIP BPE trainer against the oracle trainer (same defined tie-break): small
alphabets (many equal counts), runs, random corpus lengths and merge counts.  Usage: python scripts/dev_fuzz_trainer.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from ecg_byte_amd import rust_bpe, trainer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    alpha = [b"ab", b"abc", b"abcdefgh", b"abcdefghijklmnopqrstuvwxyz"][rng.integers(4)]
    n = int(rng.choice([2, 3, 10, 100, 1000, 20000, 150000]))
    p_run = rng.choice([0.0, 0.5, 0.9])
    parts, total = [], 0
    while total < n:
        r = int(rng.integers(1, 40)) if rng.random() < p_run else 1
        parts.append(chr(alpha[rng.integers(len(alpha))]) * r); total += r
    text = "".join(parts)[:n]
    k = int(rng.choice([1, 2, 5, 30, 200]))
    trainer.set_train_grid(int(rng.choice([0, 0, 1, 2, 3, 7, 64])))      # few workgroups: ranges of many tiles, the chains carried across tiles and ranges
    trainer.set_train_form(int(rng.choice([0, 0, 1, 2])))                 # slotted ranges with 16-bit ids (the default), with 32-bit ids, round 4's two passes
    trainer.set_train_fused(bool(rng.integers(2)))                         # the next merge's row maxima inside the merge's launch, or a launch of their own
    ids, vocab, merges = rust_bpe.byte_pair_encoding(text, k, 2)
    oids, ovocab, omerges = O.byte_pair_encoding(text, k, fast=True)
    if list(ids) != list(oids) or merges != omerges or vocab != ovocab:
        print("MISMATCH: alphabet", alpha, "n", n, "merges", k, "first difference at merge",
              next((i for i, (x, y) in enumerate(zip(merges, omerges)) if x != y), None))
        open("gpurun_out/fuzz_trainer_text.txt", "w").write(text)
        raise SystemExit(1)
    cases += 1
trainer.set_train_grid(0); trainer.set_train_form(0); trainer.set_train_fused(False)
print(f"trainer fuzz ok: {cases} corpora in {time.time() - t0:.0f} s")
