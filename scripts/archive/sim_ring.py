"""Event simulation of the claim-merge parse with a RING of staged pieces: lanes take chunk tasks in order and run ahead of the
slowest lane by up to the ring length; a piece retires (resolve + emit + restage) when no lane is left inside it."""
import os, sys, pickle, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O
import importlib.util
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "ecg-byte_amd", "synth.py")); synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
G = os.path.join(ROOT, "tests", "golden")
vocab, merges = pickle.load(open(os.path.join(G, "tokenizer_c2.pkl"), "rb"))
pct = json.load(open(os.path.join(G, "percentiles_c2.json")))
sig = synth.synth_ecg(2, 5000, seed=0, fs=500)
children = [dict()]; token = [None]
def insert(bs, tid):
    n = 0
    for b in bs:
        nx = children[n].get(b)
        if nx is None:
            nx = len(children); children.append({}); token.append(None); children[n][b] = nx
        n = nx
    token[n] = tid
for b in range(256): insert([b], b)
for bs, tid in merges: insert(bs, tid)

def simulate(sym, PIECE, P, CH, LANES=64, fuse=False, margin=208):
    """ring of P pieces of PIECE symbols; chunk starts every CH symbols; returns (iterations, lane-trips, retires)"""
    n = len(sym)
    n_pieces = (n + PIECE - 1) // PIECE
    claimed = set(); nxt_map = {}
    lanes = [None] * LANES
    oldest = 0                      # oldest piece not yet retired
    staged = min(P, n_pieces)       # pieces [oldest, staged) are in the ring
    tasks = list(range(0, n, CH)); nt = 0
    it = 0; util = 0; retires = 0; carry = 0
    while oldest < n_pieces:
        assert it < 200000
        staged_end = min(staged * PIECE, n)
        # assign tasks inside the staged window
        for l in range(LANES):
            while lanes[l] is None and nt < len(tasks) and tasks[nt] < staged_end:
                r = tasks[nt]; nt += 1
                if r in claimed: continue
                claimed.add(r); lanes[l] = [r, r, 0, r]
        retire_end = min((oldest + 1) * PIECE, n)
        pending_tasks = nt < len(tasks) and tasks[nt] < retire_end
        inside = any(st is not None and st[0] < retire_end for st in lanes)
        if not inside and not pending_tasks:
            # retire: follow the real chain through the piece
            p = carry
            while p < retire_end: p = nxt_map[p]
            carry = p
            oldest += 1; retires += 1
            if staged < n_pieces: staged += 1
            continue
        it += 1
        for l in range(LANES):
            st = lanes[l]
            if st is None: continue
            r, j, node, best = st
            if r >= staged_end: continue                   # paused at the window end: a token may start only inside the staged window
            # (a walk may read the look-ahead margin staged past the window end, as the kernel's segments do)
            util += 1
            if node != 0 and token[node] is not None: best = j
            if j - r >= 2 and sym[j - 1] == sym[j - 2] and j < n and sym[j] == sym[j - 1] and sym[j] in children[node]:
                c = sym[j]; m = 0
                while j < n and sym[j] == c and c in children[node] and m < 32:
                    if m > 0 and token[node] is not None: best = j
                    node = children[node][c]; j += 1; m += 1
                st[1] = j; st[2] = node; st[3] = best
            elif j < n and sym[j] in children[node]:
                node = children[node][sym[j]]; j += 1
                st[1] = j; st[2] = node; st[3] = best
            else:
                ln = max(best - r, 1); nxt_map[r] = r + ln
                r2 = r + ln
                if r2 >= n or r2 in claimed: lanes[l] = None
                else:
                    claimed.add(r2); lanes[l] = [r2, r2, 0, r2]
                    if fuse and r2 == j and r2 < staged_end and sym[j] in children[0]:
                        lanes[l] = [r2, j + 1, children[0][sym[j]], r2]
    return it, util, retires

if __name__ == "__main__":
    rec = int(os.environ.get("REC", "0"))
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym = (np.asarray(s).reshape(-1).astype(np.int64) + 97).tolist()
    for total, CH in ((3456, 54), (5400, 54), (5400, 84)):
        for P in (1, 2, 3, 4, 6):
            PIECE = total // P
            for fuse in (False, True):
                it, util, ret = simulate(sym, PIECE, P, CH, fuse=fuse)
                print(f"ring {total} P {P} piece {PIECE} CH {CH} fuse {int(fuse)}: iterations {it} lane-trips {util} util {util / (it * 64):.2f} retires {ret}")
