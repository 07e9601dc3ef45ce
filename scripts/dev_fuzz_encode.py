"""Dev-only: randomised parity soak of the encode kernels against the CPU oracle (byte streams and float64 records; random
vocabularies incl. long same-class chains, duplicate expansions and bytes outside the merges; lengths across segment
boundaries; every launch plan).  Usage: python scripts/dev_fuzz_encode.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import random_merges
from oracle import oracle as O
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0; streams = 0
alphabets = [b"abcdefghijklmnopqrstuvwxyz", b"abc", b"ab", b"ghij", b"abcxyz.,", b"mnopq\x80\xff", b"a",
             bytes(range(40, 100)), bytes(range(256))]          # (over 29 byte values: the general form, edge table in HBM)
while time.time() - t0 < budget:
    alpha = alphabets[rng.integers(len(alphabets))]
    max_len = int(rng.choice([3, 6, 12, 40, 120, 220]))
    n_merges = int(rng.integers(1, min(1500, 45000 // max_len)))      # (the handle takes tries below 65 535 nodes)
    merges = random_merges(rng, n_merges, alphabet=alpha, max_len=max_len, dup_frac=0.05)
    if rng.random() < 0.5:      # long runs of one symbol as expansions: same-class chains with sparse tokens
        base = len(merges)
        for k in range(int(rng.integers(1, 30))):
            ln = int(rng.integers(2, min(max_len, 200) + 1))
            merges.append(([int(alpha[rng.integers(len(alpha))])] * ln, 256 + base + k))
    tk = HipTokenizer(merges)
    trie = O.Trie(merges)
    pool = np.frombuffer(alpha + (b"~Z" if rng.random() < 0.5 else b""), dtype=np.uint8)
    B = int(rng.choice([1, 2, 5, 33, 600]))
    n = int(rng.choice([1, 7, 63, 64, 65, 1000, 3455, 3456, 3457, 4097, 9000, 20011, 70001]))
    if B * n > 8_000_000: B = max(1, 8_000_000 // n)
    # runs of random length make the run steps and the chunk balancing work
    texts = []
    for b in range(B):
        parts, total = [], 0
        p_run = rng.choice([0.0, 0.5, 0.9])
        while total < n:
            r = int(rng.integers(1, 90)) if rng.random() < p_run else 1
            parts.append(bytes([pool[rng.integers(pool.size)]]) * r); total += r
        texts.append(b"".join(parts)[:n])
    t = torch.from_numpy(np.frombuffer(b"".join(texts), dtype=np.uint8).reshape(B, n).copy()).cuda()
    plan = int(rng.integers(0, 4))
    set_encode_plan(plan)
    ids, counts = tk.encode_bytes(t)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    for b in range(B):
        want = trie.encode_bytes(texts[b])
        if counts[b] != want.size or not np.array_equal(ids[b, :counts[b]].astype(np.uint32), want):
            print("MISMATCH: alphabet", alpha, "merges", n_merges, "max_len", max_len, "B", B, "n", n, "plan", plan, "record", b)
            np.save("gpurun_out/fuzz_fail_text.npy", np.frombuffer(texts[b], dtype=np.uint8))
            import pickle; pickle.dump(merges, open("gpurun_out/fuzz_fail_merges.pkl", "wb"))
            raise SystemExit(1)
    cases += 1; streams += B
set_encode_plan(0)
print(f"fuzz ok: {cases} vocabularies, {streams} streams in {time.time() - t0:.0f} s")
