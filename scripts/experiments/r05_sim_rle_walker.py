"""Round 5, encode: trip model of a long-chunk walker (one wave = one record, lane k parses chunk k of the WHOLE record -- about 938 symbols --
from a per-lane FIFO of run-length entries, running on into the next chunk until it meets that lane's parse).  Step model = encode_flow_kernel's
(a trip is one trie step -- branch, head-into-chain, or up to 32 symbols of a same-class chain -- or the emission of a token), on the C2 tokenizer
and the bench's synthetic records.  Prints per record: real-chain steps, trips (= the slowest lane), lane utilisation, how far the overruns go,
and the back-up distance (symbols / run boundaries between a failing node and its best token) over the whole trie."""
import os, sys, pickle, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from ecg_byte_amd import synth
G = os.path.join(ROOT, "tests", "golden")
TAG = os.environ.get("TAG", "c2")
vocab, merges = pickle.load(open(os.path.join(G, f"tokenizer_{TAG}.pkl"), "rb"))
pct = json.load(open(os.path.join(G, f"percentiles_{TAG}.json")))
L = 5000 if TAG == "c2" else 1000
NREC = int(os.environ.get("NREC", "8"))
sig = synth.synth_ecg(NREC, L, seed=0)
children = [dict()]; token = [None]; depth = [0]; parent = [-1]; sym_in = [None]
def insert(bs, tid):
    n = 0
    for b in bs:
        nx = children[n].get(b)
        if nx is None:
            nx = len(children); children.append({}); token.append(None); depth.append(depth[n] + 1); parent.append(n); sym_in.append(b); children[n][b] = nx
        n = nx
    token[n] = tid
for b in range(256): insert([b], b)
for bs, tid in merges: insert(bs, tid)
# back-up distance: from every node to its deepest token-carrying ancestor-or-self, in symbols and in run boundaries crossed
best_d = [0] * len(children)
back_sym = 0; back_runs = 0
for n in range(1, len(children)):
    if token[n] is not None: best_d[n] = depth[n]
    else: best_d[n] = best_d[parent[n]]
    b = depth[n] - best_d[n]
    if b:
        # run boundaries inside the last b symbols of the path
        k = n; rb = 0
        for _ in range(b):
            p = parent[k]
            if p > 0 and sym_in[p] != sym_in[k]: rb += 1
            k = p
        back_runs = max(back_runs, rb)
    back_sym = max(back_sym, b)
print(f"trie: {len(children)} nodes, max depth {max(depth)}, max back-up {back_sym} symbols, crossing at most {back_runs} run boundaries")

def parse(sym, start, end, stop=None):
    """Greedy parse from `start` while the token start is < end (and not in `stop`).  Returns [(start, trips)], exit."""
    n = len(sym); r = start; out = []
    while r < end:
        if stop is not None and r in stop and r != start: break
        node = children[0][sym[r]]; j = r + 1; best = j if token[node] is not None else r; trips = 1   # (the emission trip; the first symbol is free)
        while j < n and sym[j] in children[node]:
            c = sym[j]
            if c == sym[j - 1] and j - r >= 2:          # inside a same-class chain: up to 32 symbols in one trip
                m = 0
                while j < n and sym[j] == c and c in children[node] and m < 32:
                    node = children[node][c]; j += 1; m += 1
                    if token[node] is not None: best = j
            else:
                node = children[node][c]; j += 1
                if token[node] is not None: best = j
            trips += 1
        out.append((r, trips))
        r += max(best - r, 1)
    return out, r

tot_trips = []; 
for rec in range(NREC):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym = (np.asarray(s).reshape(-1).astype(np.int64) + 97).tolist()
    n = len(sym)
    real, _ = parse(sym, 0, n)
    real_trips = sum(t for _, t in real)
    # chunk starts balanced by runs
    run_starts = [0] + [k for k in range(1, n) if sym[k] != sym[k - 1]]
    R = len(run_starts)
    for NL in (64,):
        starts = [run_starts[(k * R) // NL] for k in range(NL)] + [n]
        par = [parse(sym, starts[k], starts[k + 1]) for k in range(NL)]
        lane = []; over = []; joined_real = set(p for p, _ in real)
        for k in range(NL):
            toks, ex = par[k]
            t = sum(x for _, x in toks)
            kk = k + 1; r = ex; extra = 0
            while kk < NL:                                # run on until it meets the parse of a later lane
                nxt = set(p for p, _ in par[kk][0])
                while r < starts[kk + 1] and r not in nxt:
                    tt, r = parse(sym, r, r + 1)
                    extra += tt[0][1]
                if r < starts[kk + 1]: break
                kk += 1
            over.append(r - starts[k + 1] if k + 1 < NL else 0)
            lane.append(t + extra)
        lane = np.array(lane); over = np.array(over)
        tot_trips.append(lane.max())
        print(f"rec {rec}: {len(real)} tokens, {R} runs, real chain {real_trips} trips ({real_trips / len(real):.2f}/token); {NL} lanes: slowest {lane.max()} trips, mean {lane.mean():.0f}, "
              f"utilisation {lane.sum() / (NL * lane.max()):.2f}, useful {real_trips / (NL * lane.max()):.2f}; overrun into the next chunk: mean {over.mean():.0f}, p95 {np.percentile(over, 95):.0f}, max {over.max()} symbols")
print("mean trips per record:", np.mean(tot_trips), "(encode_flow_kernel: 1 178)")

# ---- second model: B1 (every lane parses its own chunk to its end) then B2 (all lanes run on together until they meet the owner's token starts)
print("\nB1 / B2 split (no claims during the own-chunk parse; joins resolved in a second phase against the owner's list):")
tt = []
for rec in range(NREC):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym = (np.asarray(s).reshape(-1).astype(np.int64) + 97).tolist()
    n = len(sym)
    run_starts = [0] + [k for k in range(1, n) if sym[k] != sym[k - 1]]
    R = len(run_starts); NL = 64
    starts = [run_starts[(k * R) // NL] for k in range(NL)] + [n]
    par = [parse(sym, starts[k], starts[k + 1]) for k in range(NL)]
    b1 = np.array([sum(x for _, x in par[k][0]) for k in range(NL)])
    b2 = []; ntok2 = []; backs = 0; cross = 0; emits = 0
    for k in range(NL):
        r = par[k][1]; kk = k + 1; extra = 0; nt = 0
        while kk < NL:
            nxt = set(p for p, _ in par[kk][0])
            while r < starts[kk + 1] and r not in nxt:
                tt_, r = parse(sym, r, r + 1); extra += tt_[0][1]; nt += 1
            if r < starts[kk + 1]: break
            kk += 1
        b2.append(extra); ntok2.append(nt)
    b2 = np.array(b2)
    tt.append(b1.max() + b2.max())
    print(f"rec {rec}: B1 slowest {b1.max()} (mean {b1.mean():.0f}), B2 slowest {b2.max()} (mean {b2.mean():.1f}, tokens max {max(ntok2)}), total {b1.max() + b2.max()} trips")
print("mean trips per record, split phases:", np.mean(tt))
