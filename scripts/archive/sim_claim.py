"""Event simulation of the claim-merge parse: 64 lanes in lockstep, one step or one emit per iteration, dynamic chunk tasks."""
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

LOG = []
FUSE = int(os.environ.get('FUSE', '0'))
def simulate(sym, SEG, CH, LANES=64, order="seq", runskip=False):
    n = len(sym); total_iters = 0; carry = 0; util = 0
    for sb in range(0, n, SEG):
        se = min(sb + SEG, n)
        claimed = set()
        if order == "runs":     # chunk starts at equal counts of run starts (D bits)
            rs = [p for p in range(sb, se) if p == sb or sym[p] != sym[p - 1]]
            R = len(rs)
            starts = sorted(set([sb] + [rs[(k * R) // LANES] for k in range(1, LANES)]))
            tasks = [max(s, carry) for s in starts]
        else:
            tasks = [max(s, carry) for s in range(sb, se, CH) if s + CH > carry or s >= carry]
        tasks = [t for i, t in enumerate(tasks) if i == 0 or t != tasks[i - 1]]
        if order == "interleave":   # lanes' initial tasks spread out: 0, 2, 4, ... then odd ones
            tasks = tasks[0::2] + tasks[1::2]
        nxt = 0
        lanes = [None] * LANES   # state: [r, j, node, best]
        it = 0
        nxt_map = {}
        busy = [0] * LANES; first = [None] * LANES; last = [0] * LANES; ntok = [0] * LANES
        steps_kind = [0, 0, 0, 0]   # branch/cont, run, emit, claimfail
        while True:
            # assign tasks
            for l in range(LANES):
                while lanes[l] is None and nxt < len(tasks):
                    r = tasks[nxt]; nxt += 1
                    if r in claimed or r >= se: continue
                    claimed.add(r); lanes[l] = [r, r, 0, r]; first[l] = r
            if all(x is None for x in lanes): break
            it += 1
            for l in range(LANES):
                st = lanes[l]
                if st is None: continue
                util += 1; busy[l] += 1
                r, j, node, best = st
                if node != 0 and token[node] is not None: best = j
                if runskip and j - r >= 2 and sym[j - 1] == sym[j - 2] and j < n and sym[j] == sym[j - 1] and sym[j] in children[node]:
                    c = sym[j]; m = 0
                    while j < n and sym[j] == c and c in children[node] and m < 32:
                        if m > 0 and token[node] is not None: best = j
                        node = children[node][c]; j += 1; m += 1
                    st[1] = j; st[2] = node; st[3] = best
                elif j < n and sym[j] in children[node]:
                    node = children[node][sym[j]]; j += 1
                    st[1] = j; st[2] = node; st[3] = best
                else:
                    ln = max(best - r, 1); nxt_map[r] = r + ln; ntok[l] += 1; last[l] = r + ln
                    r2 = r + ln
                    if r2 >= se or r2 in claimed: lanes[l] = None
                    else:
                        claimed.add(r2); lanes[l] = [r2, r2, 0, r2]
                        if FUSE and r2 == j and j < n and sym[j] in children[0]:      # the failed symbol is the next token's first
                            lanes[l] = [r2, j + 1, children[0][sym[j]], r2]
        total_iters += it
        LOG.append((sb, it, list(busy), list(steps_kind), list(first), list(last), list(ntok)))
        # true chain
        p = carry
        while p < se: p = nxt_map[p]
        carry = p
    return total_iters, util

for rec in range(2):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym = (np.asarray(s).reshape(-1).astype(np.int64) + 97).tolist()
    for SEG in (3456, 4096):
        for order in ("seq", "runs"):
            LOG.clear()
            it, util = simulate(sym, SEG, SEG // 64, runskip=True, order=order)
            print("rec", rec, "SEG", SEG, order, "fuse", FUSE, "iterations", it, "util %.2f" % (util / (it * 64)))
