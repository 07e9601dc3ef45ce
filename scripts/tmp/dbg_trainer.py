import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from ecg_byte_amd import rust_bpe, trainer
text = b"a" * 4095
for form in (0, 1):
    trainer.set_train_form(form)
    for nm in range(1, 15):
        got = rust_bpe.byte_pair_encoding(text, nm, 1)
        want = O.byte_pair_encoding(text, nm, fast=True)
        ok = got == want
        print(form, nm, ok, len(got[0]), len(want[0]))
        if not ok:
            print(" got ", got[0][:40], got[2][-2:])
            print(" want", want[0][:40], want[2][-2:])
            break
