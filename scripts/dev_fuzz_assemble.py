"""Dev-only: randomised parity soak of the sequence-assembly kernel against the oracle's restatement of _prepare_training /
_prepare_inference: random signal lengths around pad_to_max (truncate / pad / exact), random Q / A lengths incl. empty, pad
ids inside Q.  Usage: python scripts/dev_fuzz_assemble.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import assemble as OA
from ecg_byte_amd.data_loader import BatchAssembler

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
pad, bos, eos, s0, s1 = 132014, 128000, 128001, 132012, 132013
lut = (np.arange(0, 4400, dtype=np.int32) + 128256)
t0 = time.time(); cases = 0; rows = 0
while time.time() - t0 < budget:
    P = int(rng.choice([8, 31, 64, 100, 500, 1020]))
    asm = BatchAssembler([([97, 98], 256)], lut, pad, bos, eos, s0, s1, P)
    B = int(rng.choice([1, 2, 7, 40]))
    nmax = P + 40
    counts = rng.integers(0, nmax + 1, size=B)
    ids = rng.integers(0, 4400, size=(B, max(1, nmax)))
    qs = [rng.integers(1000, 100000, size=int(rng.integers(0, max(1, min(25, P // 3))))).tolist() for _ in range(B)]
    ans = [rng.integers(1000, 100000, size=int(rng.integers(0, max(1, min(33, P // 3))))).tolist() for _ in range(B)]
    for q in qs:
        if q and rng.random() < 0.2: q[rng.integers(len(q))] = pad
    try:
        out = asm.assemble(torch.from_numpy(ids.astype(np.int32)).cuda(), torch.from_numpy(counts.astype(np.int32)).cuda(), qs, ans)
    except AssertionError:
        continue                                              # Q + A longer than the row: the host refuses, as documented
    for b in range(B):
        r = OA.prepare_training(lut[ids[b, :counts[b]]].tolist(), qs[b], ans[b], pad, bos, eos, s0, s1, P)
        for k in r:
            if not np.array_equal(out[k][b].cpu().numpy(), r[k]):
                print("MISMATCH", "P", P, "B", B, "row", b, "key", k, "count", counts[b], "q", len(qs[b]), "a", len(ans[b]))
                raise SystemExit(1)
    cases += 1; rows += B
print(f"assemble fuzz ok: {cases} batches, {rows} rows in {time.time() - t0:.0f} s")
