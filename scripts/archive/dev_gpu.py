import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, _ROOT); sys.path.insert(0, _ROOT + '/tests')
import numpy as np, torch
from helpers import load_tokenizer, oracle_batch
from ecg_byte_amd import synth
from ecg_byte_amd.tokenizer import HipTokenizer, quantize
from oracle import oracle as O
for tag, L, B in (("c1",1000,8),("c2",5000,8),("c2",5000,1100)):
    vocab, merges, pc = load_tokenizer(tag)
    tk = HipTokenizer(merges); print(tag, "nodes", tk.n_nodes)
    x = synth.synth_ecg(min(B,64), L, seed=0)
    if B > 64: x = np.concatenate([x]*((B+63)//64))[:B]
    xd = torch.from_numpy(x).cuda()
    sym = quantize(xd, pc).cpu().numpy()
    print(" quantize ok:", np.array_equal(sym, O.quantize(x, pc['percentile_1'], pc['percentile_99'])))
    ids, counts = tk.quantize_encode(xd, pc)
    torch.cuda.synchronize()
    ids = ids.cpu().numpy(); counts = counts.cpu().numpy()
    tr = O.Trie(merges)
    ref = oracle_batch(tr, x[:64], pc)
    ok = all(counts[b]==len(ref[b%64]) and np.array_equal(ids[b,:counts[b]], ref[b%64]) for b in range(B))
    print(" encode ok:", ok, counts[:4], [len(r) for r in ref[:4]])
    for it in range(3):
        torch.cuda.synchronize(); t=time.time()
        tk.quantize_encode(xd, pc); torch.cuda.synchronize(); dt=time.time()-t
        print("  B",B,"time ms", dt*1e3, "ECG/s", B/dt, "GB/s", B*12*L*8/dt/1e9)
