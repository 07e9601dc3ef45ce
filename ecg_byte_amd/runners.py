"""Mirrors of ecg_byte/runners/train.py (`trainer`, `validater`) and ecg_byte/runners/inference.py (`tester`):
same arguments, same return dicts, same checkpoint files.  Differences, all on the device side: the gradient
clip of train.py:26 is folded into `optimizer.step_and_update_lr()` (decoder.HipAdam applies the global-norm
clip to 1.0 inside the Adam kernel), and in a multi-rank run the gradients are exchanged by
`parallel.GradAllReduce` during backward instead of DDP hooks.  wandb logging is not mirrored."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .model_utils import evaluate_strings


def _save(checkpoint, path, args):
    if getattr(args, "dis", False):
        dist.barrier()
        if dist.get_rank() == 0:
            torch.save(checkpoint, path)
    else:
        torch.save(checkpoint, path)


def trainer(model, dataloader, optimizer, args, epoch, directory_path, checkpoint_every=50000):
    """train.py:7-73"""
    model.train()
    if getattr(args, "dis", False) and hasattr(getattr(dataloader, "sampler", None), "set_epoch"):
        dataloader.sampler.set_epoch(epoch)
    total_loss, len_of_batch, dev_count = 0.0, 0, 0
    for step, batch in enumerate(dataloader):
        if batch is None:
            print(f"Skipping invalid batch at step {step}")
            continue
        try:
            optimizer.zero_grad()
            out = model(batch)
            loss = out.loss
            loss.backward()
            optimizer.step_and_update_lr()          # clip_grad_norm_(1.0) happens inside
            total_loss += loss.item()
            len_of_batch += 1
            if ((step + 1) % checkpoint_every == 0) and not getattr(args, "toy", False):
                _save({"model": model.state_dict(), "epoch": epoch}, f"{directory_path}/best_train_model_{epoch}_{step}.pth", args)
                print(f"Best model saved at epoch: {epoch+1} {step}")
            if getattr(args, "dev", False):
                dev_count += 1
                if dev_count == 10:
                    break
        except Exception as e:
            print(f"Error during training at step {step}: {e}")
            if getattr(args, "dis", False):
                # train.py:59-61 swallows the error on every rank alike; with a collective inside backward a rank that skips a
                # step leaves the others waiting in all-reduce forever, so a multi-rank run stops instead
                raise
            continue
    if len_of_batch == 0:
        print("No valid batches for training.")
        return {"average_loss": float("inf")}
    return {"average_loss": total_loss / len_of_batch}


def validater(model, dataloader, args, epoch):
    """train.py:75-117.  The loss of a batch is computed with the forward-only path (no activations kept)."""
    model.eval()
    total_loss, len_of_batch, dev_count = 0.0, 0, 0
    with torch.no_grad():
        for step, batch in enumerate(dataloader):
            if batch is None:
                print(f"Skipping invalid batch at step {step}")
                continue
            try:
                out = model(batch)
                total_loss += out.loss.item()
                len_of_batch += 1
                if getattr(args, "dev", False):
                    dev_count += 1
                    if dev_count == 10:
                        break
            except Exception as e:
                print(f"Error during validation at step {step}: {e}")
                continue
    if len_of_batch == 0:
        print("No valid batches for validation.")
        return {"average_loss": float("inf")}
    return {"average_loss": total_loss / len_of_batch}


def tester(model, dataloader, tokenizer, args, extra_metrics=None):
    """inference.py:7-78: batch-1 greedy generation, per-sample metrics, averages over the samples."""
    model.eval()
    all_results, gt_answers, gen_answers, questions = [], [], [], []
    dev_count = 0
    with torch.no_grad():
        for batch in dataloader:
            if batch is None:
                print("Skipping invalid batch ")
                continue
            answer = batch["answer"]
            out = None
            try:
                out = [model.generate(batch, tokenizer)]
                all_results.append(evaluate_strings(answer, out, getattr(args, "device", None), extra_metrics))
                gt_answers.append(answer[0])
                gen_answers.append(out[0])
                questions.append(batch["question"][0])
            except Exception as e:
                print("could not evaluate for some reason:", str(e))
                print(f"Error type: {type(e).__name__}")
                # inference.py:34-36: a failed sample scores zero in EVERY metric, so it pulls all the averages down alike
                zero = {"BLEU": 0}
                if extra_metrics is not None:
                    zero.update({"METEOR": 0, "ROUGE": {"rouge-1": 0, "rouge-2": 0, "rouge-l": 0},
                                 "BERTSCORE": {"hf-prec": [0], "hf-rec": [0], "hf-f1": [0]}})
                all_results.append(zero)
            if getattr(args, "dev", False):
                dev_count += 1
                if dev_count == 10:
                    break
    sums, counts = {}, {}
    for entry in all_results:
        for key, value in entry.items():
            if isinstance(value, dict):           # ROUGE / BERTSCORE style sub-dicts (inference.py:57-63)
                for sub_key, sub_value in value.items():
                    v = sub_value[0] if isinstance(sub_value, (list, tuple)) else sub_value
                    sums[sub_key] = sums.get(sub_key, 0) + v
                    counts[sub_key] = counts.get(sub_key, 0) + 1
            else:
                sums[key] = sums.get(key, 0) + value
                counts[key] = counts.get(key, 0) + 1
    return {"metrics": {k: sums[k] / counts[k] for k in sums},
            "qa_results": {"questions": questions, "gt_answers": gt_answers, "gen_answers": gen_answers}}
