"""Dev-only: the tokenizer trainer on a corpus of the C2 shape (1.2e8 symbols, 4 000 merges) -- a quantised random walk, or with CORPUS=c2 the bench leg's own corpus -- for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ecg_byte_amd import _lib
if os.environ.get("ECGB_SO"): _lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), os.environ["ECGB_SO"])
from ecg_byte_amd.trainer import bpe_train_device
if os.environ.get("CORPUS") == "c2":                    # the bench's trainer leg: 2 000 synthetic records of seed 1, quantised with the C2 tokenizer's percentiles
    import bench
    from ecg_byte_amd.tokenizer import quantize
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import load_tokenizer
    _, _, pc = load_tokenizer("c2")
    x = bench.make_signals(2000, 5000, seed=1, start=0, workers=1)
    text = (quantize(torch.from_numpy(x).cuda(), pc).view(-1) + 97).contiguous()
    del x
else:
    rng = np.random.default_rng(1)
    n = int(os.environ.get("N", 2000 * 12 * 5000))
    # a random walk over 26 symbols: neighbouring samples differ by at most one level, like a quantised ECG
    steps = rng.integers(-1, 2, size=n, dtype=np.int8)
    sym = np.clip(np.cumsum(steps) % 52, 0, 51); sym = np.where(sym > 25, 51 - sym, sym).astype(np.uint8)
    text = torch.from_numpy(sym + 97).cuda()
from ecg_byte_amd import trainer
forms = [int(f) for f in os.environ.get("FORMS", "0").split(",")]          # ecgb_set_bpe_train_form: 0 slotted ranges + 16-bit ids, 1 slotted + 32-bit, 2 round 4's two passes
ref = None
for form in forms:
  trainer.set_train_form(form)
  for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ids, n_ids, pairs, n_done = bpe_train_device(text, 4000)
    torch.cuda.synchronize(); print(f"form {form}: {time.perf_counter() - t0:.3f} s, merges {int(n_done.item())}, ids {int(n_ids.item())}")
  got = (ids[: int(n_ids.item())].clone(), pairs.clone())
  if ref is None: ref = got
  else: print("  same ids and merges as the first form:", torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]))
trainer.set_train_form(0)
