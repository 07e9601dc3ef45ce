"""CPU simulation of the speculative chunk parse to count wave iterations under different schedules."""
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
sig = synth.synth_ecg(4, 5000, seed=0, fs=500)
# build trie
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
print("nodes", len(children), "max token len", max(len(b) for b, _ in merges))

RUNSKIP = int(os.environ.get('RUNSKIP', '0'))
RUNSTART = int(os.environ.get('RUNSTART', '0'))
def walk(sym, start, stop_at, marks=None):
    """greedy parse from start until position >= stop_at; returns (list of (r,len)), iterations, exit). iterations = steps + emits"""
    n = len(sym); r = start; its = 0; toks = []
    while r < stop_at:
        if marks is not None and marks[r]:
            return toks, its, None
        node = 0; j = r; best = r
        while True:
            its += 1
            if token[node] is not None and node != 0: best = j
            if RUNSKIP and j - r >= 2 and sym[j - 1] == sym[j - 2] and j < n and sym[j] == sym[j - 1] and sym[j] in children[node]:
                c = sym[j]; m = 0
                while j < n and sym[j] == c and c in children[node] and m < 32:
                    if m > 0 and token[node] is not None: best = j
                    node = children[node][c]; j += 1; m += 1
            elif j < n and sym[j] in children[node]:
                node = children[node][sym[j]]; j += 1
            else:
                break
        ln = max(best - r, 1)
        toks.append((r, ln, j - r))
        r += ln
    return toks, its, r

for rec in range(2):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym = (np.asarray(s).reshape(-1).astype(np.int64) + 97).tolist()
    n = len(sym)
    toks, its, _ = walk(sym, 0, n)
    lens = np.array([t[1] for t in toks]); depth = np.array([t[2] for t in toks])
    print("record", rec, "n", n, "tokens", len(toks), "serial iterations", its, "mean len %.1f max %d" % (lens.mean(), lens.max()),
          "lookahead waste", int((depth - lens).sum()), "frac symbols in tokens>=64: %.2f" % (lens[lens >= 64].sum() / n))
    true_marks = np.zeros(n + 1, bool); true_marks[[t[0] for t in toks]] = True
    for CH in (32, 64, 128):
        SEG = 64 * CH
        first_lock = 0; first_work = 0; stitch_lock = 0; stitch_work = 0; nseg = 0
        for sb in range(0, n, SEG):
            nseg += 1
            se = min(sb + SEG, n)
            per = []; exits = []; markss = []
            for c in range(64):
                s0 = sb + c * CH
                if s0 >= se: break
                e0 = min(s0 + CH, se)
                st = s0
                if RUNSTART and s0 > 0:
                    while st < e0 and sym[st] == sym[st - 1]: st += 1
                tk, it, ex = walk(sym, st, e0) if st < e0 else ([], 0, st)
                per.append(it); exits.append(ex)
                m = np.zeros(n + 1, bool); m[[t[0] for t in tk]] = True; markss.append(m)
            first_lock += max(per); first_work += sum(per)
            # stitch pass 1: lane c re-walks from true entry until it hits own speculative marks (approximate: use the true chain)
            re = []
            for c in range(len(per)):
                s0 = sb + c * CH; e0 = min(s0 + CH, se)
                # true entry = first true token start >= s0
                idx = np.nonzero(true_marks[s0:n + 1])[0]
                ent = s0 + int(idx[0]) if len(idx) else n
                if ent >= e0 or c == 0 and sb == 0: re.append(0); continue
                tk, it, ex = walk(sym, ent, e0, markss[c])
                re.append(it)
            stitch_lock += max(re); stitch_work += sum(re)
        print("  chunk", CH, "segments", nseg, "first: lockstep %d balanced %.0f | stitch(1 pass, true entries): lockstep %d balanced %.0f"
              % (first_lock, first_work / 64, stitch_lock, stitch_work / 64))
