"""Dev-only: the loss trajectory of tests/test_gpu_pipeline.py::test_runners_train_validate_checkpoint_and_generate (how far its
thresholds are from the values)."""
import os, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest, torch
import test_gpu_pipeline as T
from ecg_byte_amd.data_loader import DeviceBatchLoader
from ecg_byte_amd.llm import LLM
from ecg_byte_amd.runners import trainer, validater
import pathlib
for rep in range(3):
    with tempfile.TemporaryDirectory() as d:
        ptb = T.ptb_dir.__wrapped__(pathlib.Path(d)) if hasattr(T.ptb_dir, "__wrapped__") else None
        if ptb is None: raise SystemExit("fixture not callable")
        root, x, pc = ptb
        train_ds, tok, args = T._dataset(root, "train")
        val_ds, _, _ = T._dataset(root, "val")
        model = LLM(T._tiny_model(tok), args)
        opt = model.llm.make_optimizer(lr=1e-4, warmup=4)
        run_dir = os.path.join(d, "run"); os.makedirs(run_dir)
        losses, vals = [], []
        for epoch in range(6):
            losses.append(trainer(model, DeviceBatchLoader(train_ds, batch_size=4, shuffle=True, seed=1), opt, args, epoch, run_dir, checkpoint_every=50000)["average_loss"])
            vals.append(validater(model, DeviceBatchLoader(val_ds, batch_size=2), args, epoch)["average_loss"])
        print("losses", [round(v, 4) for v in losses], "ratio %.3f" % (losses[-1] / losses[0]), "vals", [round(v, 4) for v in vals])
