import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_decoder_model as T
zg, m = T._load_generate()
ids = torch.from_numpy(zg["input_ids"]).cuda(); mask = torch.from_numpy(zg["attention_mask"]).cuda()
S0 = ids.shape[1]
for uc in (True, False):
    seq, lg = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=24, pad_token_id=299, eos_token_id=int(zg["eos_token_id"]), use_cache=uc, return_logits=True)
    seq = seq.cpu().numpy(); lg = lg.cpu().numpy(); ref = zg["scores"]; want = zg["sequences"]
    print("use_cache", uc, "match", (seq[:, S0:] == want[:, S0:S0 + seq.shape[1] - S0]).mean(axis=1))
    for b in range(3):
        errs = []
        for t in range(lg.shape[1]):
            r = ref[b, t]
            if not np.isfinite(r).all(): break
            top2 = np.sort(r)[-2:]
            errs.append((float(np.abs(lg[b, t] - r).max()), float(r.std()), float(top2[1] - top2[0]), int(seq[b, S0 + t] == want[b, S0 + t])))
        print(b, [tuple(round(x, 3) for x in e) for e in errs[:8]])
