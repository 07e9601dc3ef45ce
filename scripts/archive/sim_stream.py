import os, sys, pickle, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from ecg_byte_amd import synth
G = os.path.join(ROOT, "tests", "golden")
vocab, merges = pickle.load(open(os.path.join(G, "tokenizer_c2.pkl"), "rb"))
pct = json.load(open(os.path.join(G, "percentiles_c2.json")))
sig = synth.synth_ecg(2, 5000, seed=0)
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

def sim(sym, RING, THRESH, KMIN, W=128, NL=64, LINE=16, maxtrips=3000, RUNTRICK=True):
    n=len(sym)
    CH=((n+NL-1)//NL+15)//16*16
    r=[min(k*CH,n) for k in range(NL)]
    j=list(r); node=[0]*NL; best=list(r)
    F=[(min(k*CH,n)//16)*16 for k in range(NL)]
    fresh=[True]*NL
    done=[r[k]>=n for k in range(NL)]
    claims=[set() for _ in range(NL)]
    windone=[done[k] for k in range(NL)]
    trips=rounds=lines=busy=stall=rewinds=waitwin=0
    nend=(n+15)//16*16
    bts=[]
    def need_lo(k):
        if fresh[k]: return r[k]
        b=max(best[k],r[k]+1) if best[k]>r[k] else r[k]+1   # next token starts at best (or r+1 for an unmatched byte)
        if RUNTRICK and j[k]>b and all(sym[q]==sym[j[k]-1] for q in range(b, j[k])): return j[k]
        return min(b,j[k])
    while not all(done) and trips<maxtrips:
        trips+=1
        want=[False]*NL; starved=[False]*NL
        for k in range(NL):
            if done[k] or F[k]>=nend: continue
            lo=need_lo(k)
            room=(F[k]+LINE-RING)<=lo
            want[k]=(F[k]-j[k])<THRESH and room
            starved[k]=j[k]>=F[k]
        if any(starved) or sum(want)>=KMIN:
            rounds+=1
            for k in range(NL):
                if want[k] or starved[k]:
                    F[k]=min(F[k]+LINE,nend); lines+=1
        for k in range(NL):
            if done[k]: continue
            if j[k]>=F[k] and j[k]<n:
                stall+=1; continue
            if need_lo(k)<F[k]-RING:          # history evicted: rewind
                rewinds+=1; F[k]=(need_lo(k)//16)*16; continue
            if fresh[k]:
                c=min(r[k]//CH,NL-1)
                if c>k:
                    if not windone[c]:
                        waitwin+=1; continue
                    if r[k] in claims[c]:
                        done[k]=True; continue
                node[k]=children[0][sym[r[k]]]; j[k]=r[k]+1; best[k]=r[k]; fresh[k]=False
                if c==k:
                    if r[k]-k*CH<W: claims[k].add(r[k])
                    else: windone[k]=True
                if j[k]>=F[k] and j[k]<n:
                    busy+=1; continue      # took the first symbol this trip, next symbol not staged
            busy+=1
            if token[node[k]] is not None: best[k]=j[k]
            jj=j[k]
            if jj<n and sym[jj] in children[node[k]]:
                c_=sym[jj]
                if sym[jj]==sym[jj-1] and jj-r[k]>=2:
                    m=0
                    while jj<n and jj<F[k] and sym[jj]==c_ and c_ in children[node[k]] and m<32:
                        if m>0 and token[node[k]] is not None: best[k]=jj
                        node[k]=children[node[k]][c_]; jj+=1; m+=1
                else:
                    node[k]=children[node[k]][c_]; jj+=1
                j[k]=jj
            else:
                ln=max(best[k]-r[k],1)
                bts.append(j[k]-(r[k]+ln))
                r[k]+=ln; j[k]=r[k]; fresh[k]=True
                if r[k]>=n: done[k]=True; windone[k]=True
                elif r[k]>=(k+1)*CH: windone[k]=True
    bt=np.array(bts)
    return dict(trips=trips, rounds=rounds, lines=lines, util=round(busy/(trips*NL),3), stall=stall, rewinds=rewinds, waitwin=waitwin,
                lpr=round(lines/max(rounds,1),1), bt_gt16=round(float((bt>16).mean()),4), bt_gt32=round(float((bt>32).mean()),4), bt_max=int(bt.max()))

t0=time.time()
for rec in range(2):
    s = O.quantize(sig[rec], pct["percentile_1"], pct["percentile_99"])
    sym=(np.asarray(s).reshape(-1).astype(np.int64)+97).tolist()
    for RING,THRESH,KMIN in ((64,32,16),(64,32,32),(64,40,24),(64,48,32),(48,32,24),(128,64,32),(128,96,48)):
        print(rec,(RING,THRESH,KMIN),sim(sym,RING,THRESH,KMIN), round(time.time()-t0,1), flush=True)
