"""Dev-only: randomised parity soak of the FUSED quantise+encode (float64 records) against the CPU oracle: random percentiles
(incl. large offsets and tiny ranges), samples on and next to the bin edges, NaN / inf / huge values, ragged record lengths,
every launch plan.  Usage: python scripts/dev_fuzz_quantize_encode.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import load_tokenizer, random_merges
from oracle import oracle as O
from ecg_byte_amd.tokenizer import HipTokenizer, set_encode_plan

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
toks = []
for tag in ("c1", "c2"):
    _, merges, _ = load_tokenizer(tag)
    toks.append((HipTokenizer(merges), O.Trie(merges)))
m3 = random_merges(rng, 300, alphabet=b"abcdefghijklmnopqrstuvwxyz", max_len=30)
toks.append((HipTokenizer(m3), O.Trie(m3)))
t0 = time.time(); cases = 0; recs = 0
while time.time() - t0 < budget:
    tk, trie = toks[rng.integers(len(toks))]
    off = float(rng.choice([0.0, 0.0, 1.0, -3.5, 1e3, -1e6, 1e9, 3e12]))
    width = float(rng.choice([1e-3, 0.37, 1.0, 2.5, 40.0, 1e4]))
    p1, p99 = off - width * rng.random(), off + width * (0.5 + rng.random())
    B = int(rng.choice([1, 3, 17, 520]))
    L = int(rng.choice([1, 2, 3, 5, 250, 999, 1000, 1001, 5000]))
    x = p1 - 0.6 + (p99 - p1 + 1.2) * rng.random((B, 12, L)) * rng.choice([1.0, 1.0, 1.3])
    # exact bin edges of the reference arithmetic and their neighbours
    a, d = p1 - 0.5, ((p99 + 0.5) - (p1 - 0.5)) + 1e-6
    edges = a + d * (np.arange(0, 27) / 26.0)
    flat = x.reshape(-1)
    k = min(flat.size, 400)
    idx = rng.choice(flat.size, size=k, replace=False)
    vals = edges[rng.integers(0, 27, size=k)]
    for _ in range(int(rng.integers(0, 4))):
        vals = np.nextafter(vals, np.where(rng.random(k) < 0.5, -np.inf, np.inf))
    flat[idx] = vals
    sp = rng.choice(flat.size, size=min(flat.size, 12), replace=False)
    flat[sp] = rng.choice([np.nan, np.inf, -np.inf, 1e300, -1e300, 0.0], size=sp.size)
    pc = {"percentile_1": p1, "percentile_99": p99}
    plan = int(rng.choice([0, 1, 2, 3, 4, 6]))                # (4: a lane per run-length chunk, 6: stager + walker waves -- round 5's matchers; they fall back to the segment kernels where they do not apply)
    set_encode_plan(plan)
    ids, counts = tk.quantize_encode(torch.from_numpy(np.ascontiguousarray(x)).cuda(), pc)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    for b in range(min(B, 40)):
        want = trie.quantize_encode(x[b], p1, p99)
        if counts[b] != want.size or not np.array_equal(ids[b, :counts[b]].astype(np.uint32), want):
            print("MISMATCH: p1", repr(p1), "p99", repr(p99), "B", B, "L", L, "plan", plan, "record", b)
            raise SystemExit(1)
    cases += 1; recs += min(B, 40)
set_encode_plan(0)
print(f"fuzz ok: {cases} batches, {recs} records checked in {time.time() - t0:.0f} s")
