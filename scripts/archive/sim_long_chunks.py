import os, sys, pickle, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from ecg_byte_amd import synth
G = os.path.join(ROOT, "tests", "golden")
vocab, merges = pickle.load(open(os.path.join(G, "tokenizer_c2.pkl"), "rb"))
pct = json.load(open(os.path.join(G, "percentiles_c2.json")))
sig = synth.synth_ecg(4, 5000, seed=0)
children=[dict()]; token=[None]
def insert(bs, tid):
    n=0
    for b in bs:
        nx=children[n].get(b)
        if nx is None:
            nx=len(children); children.append({}); token.append(None); children[n][b]=nx
        n=nx
    token[n]=tid
for b in range(256): insert([b], b)
for bs,tid in merges: insert(bs,tid)

def parse_from(sym, start, stop_set=None, end=None):
    """greedy parse from start; returns list of (token_start, steps_for_token) until reaching >= end or a start in stop_set"""
    n=len(sym); r=start; out=[]
    while r < (end if end is not None else n):
        if stop_set is not None and r in stop_set and r!=start: break
        j=r; node=0; best=r; steps=0
        # first symbol free
        node=children[0][sym[j]]; j+=1
        while True:
            if token[node] is not None: best=j
            steps+=1
            if j<n and sym[j] in children[node]:
                c=sym[j]
                # run step: if the symbol repeats previous and node in chain
                if sym[j]==sym[j-1] and j-r>=2:
                    m=0
                    while j<n and sym[j]==c and c in children[node] and m<32:
                        if m>0 and token[node] is not None: best=j
                        node=children[node][c]; j+=1; m+=1
                else:
                    node=children[node][c]; j+=1
            else:
                break
        ln=max(best-r,1)
        out.append((r,steps))
        r+=ln
    return out, r

for rec in range(4):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym=(np.asarray(s).reshape(-1).astype(np.int64)+97).tolist()
    n=len(sym)
    real,_=parse_from(sym,0)
    tot=sum(st for _,st in real)
    print("rec",rec,"tokens",len(real),"real steps",tot)
    for NCH in (64,128,256):
        CH=(n+NCH-1)//NCH
        starts=[k*CH for k in range(NCH)]
        # each chunk parse from its start to the end of the chunk; then continue until joined with next chunk's parse
        lane_steps=[]; waste=0
        parses=[]
        for k in range(NCH):
            toks,exit_=parse_from(sym,starts[k],end=min(n,starts[k]+CH))
            parses.append((toks,exit_))
        for k in range(NCH):
            toks,exit_=parses[k]
            st=sum(x for _,x in toks)
            # overrun into chunk k+1 until join
            if k+1<NCH:
                nxt=set(p for p,_ in parses[k+1][0])
                r=exit_
                extra=0
                while r<n and r not in nxt:
                    t,r2=parse_from(sym,r,end=r+1)
                    extra+=t[0][1]; r=r2
                st+=extra; waste+=extra
            lane_steps.append(st)
        ls=np.array(lane_steps)
        # static: 64 lanes, chunks assigned round-robin in groups (NCH/64 chunks per lane sequential)
        per_lane=ls.reshape(64,-1).sum(1) if NCH%64==0 else None
        print("  NCH",NCH,"sum",ls.sum(),"waste",waste,"max chunk",ls.max(),"mean",ls.mean(), "static contiguous lanes: max",per_lane.max(),"util %.2f"%(per_lane.sum()/(64*per_lane.max())))
        # interleaved assignment: lane l takes chunks l, l+64, ...
        pl=ls.reshape(-1,64).sum(0)
        print("     interleaved: max",pl.max(),"util %.2f"%(pl.sum()/(64*pl.max())))
        # dynamic grabbing (greedy list scheduling in chunk order)
        import heapq
        h=[0]*64; heapq.heapify(h)
        for x in ls:
            t=heapq.heappop(h); heapq.heappush(h,t+x)
        mx=max(h); print("     dynamic: max",mx,"util %.2f"%(ls.sum()/(64*mx)))
