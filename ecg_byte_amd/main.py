"""Mirror of ecg_byte/main.py (the trainer / inference CLI of the end-to-end path): same options, same
directory conventions, same checkpoint and result files, with the hot path on the MI355X:

    python -m ecg_byte_amd.main --dis --gpus 0,1,2,3 --ports 12359 --model <dir> --dataset ptb_500 ...      (the reference's launch)
    python -m torch.distributed.run --nproc-per-node N -m ecg_byte_amd.main --dis --model <dir> --dataset ptb_500 ...
    python -m ecg_byte_amd.main --device cuda:0 --model <dir> --dataset ptb_500 --tokenizer_check tokenizer_3500_300000 ...

Differences from the reference, all deliberate:
  * `--model` is a LOCAL directory in the hub layout (config.json, model.safetensors, tokenizer files); nothing is
    downloaded and no API key is read (main.py:82-87 logs in to the hub);
  * distributed runs are one process per GPU over RCCL, started EITHER way: by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment), or -- with no RANK in the environment -- by this module itself exactly as the reference does it
    (`--dis --gpus 0,1,2,3 --ports P`: mp.spawn of one child per listed GPU before any GPU call, main.py:57-63, 356-360), so
    scripts/train_model.sh runs with only the module name changed;
  * `--data_root` / `--runs_root` (defaults: the reference's ./data and ./runs) say where the files live;
  * the decoder is decoder.HipCausalLM (Llama family only), LoRA is decoder.LoraSite, the optimizer decoder.HipAdam
    (Adam + L2 + Noam schedule + clip 1.0 in one kernel), batches come from data_loader.DeviceBatchLoader;
  * wandb logging (--log) and the attention visualisation (--interpret) are not built."""
from __future__ import annotations

import argparse
import gc
import json
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def get_args(argv=None):
    """main.py:26-55, plus --data_root / --runs_root / --workers."""
    p = argparse.ArgumentParser(description=None)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--epochs", type=int, default=150)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--dataset", type=str, default="mimic_500")
    p.add_argument("--model", type=str, default=None)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.99)
    p.add_argument("--eps", type=float, default=1e-8)
    p.add_argument("--warmup", type=int, default=500)
    p.add_argument("--weight_decay", type=float, default=1e-2)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--patience", type=int, default=5)
    p.add_argument("--dev", action="store_true")
    p.add_argument("--inference", action="store_true")
    p.add_argument("--checkpoint", type=str)
    p.add_argument("--log", action="store_true")
    p.add_argument("--dis", action="store_true")
    p.add_argument("--tokenizer_check", type=str)
    p.add_argument("--num_merges", type=int, default=1000)
    p.add_argument("--pad_to_max", type=int, default=1000)
    p.add_argument("--gpus", type=str, default="0")
    p.add_argument("--ports", type=str, default="12355")
    p.add_argument("--toy", action="store_true")
    p.add_argument("--peft", action="store_true", default=None)
    p.add_argument("--percentiles", type=str, default=None)
    p.add_argument("--interpret", action="store_true")
    p.add_argument("--data_root", type=str, default="./data")
    p.add_argument("--runs_root", type=str, default="./runs")
    p.add_argument("--workers", type=int, default=4, help="reader threads of the batch loader")
    return p.parse_args(argv)


def run_directory(args):
    """main.py:98"""
    return (f"{args.runs_root}/{args.seed}/{args.model}_{args.dataset}_{args.lr}_{args.beta1}_{args.beta2}_{args.eps}_"
            f"{args.weight_decay}_{args.warmup}_{args.batch_size}_{args.epochs}_{args.num_merges}_{args.pad_to_max}_{args.toy}")


def build_model_and_tokenizer(args, vocab, device):
    """main.py:140-158: tokenizer + `signal_{id}` tokens in pickled-dict order, <sig_start>, <sig_end>, <pad>; embeddings
    resized; optional LoRA; LLM wrapper."""
    from transformers import AutoTokenizer            # tokenizer plumbing only (site-packages, not the vendored copy)
    from .decoder import HipCausalLM
    from .llm import LLM
    tokenizer = AutoTokenizer.from_pretrained(args.model)
    llm = HipCausalLM.from_pretrained(args.model, device=device)
    tokenizer.add_tokens([f"signal_{str(ids)}" for ids in vocab.keys()])
    tokenizer.add_tokens(["<sig_start>"], special_tokens=True)
    tokenizer.add_tokens(["<sig_end>"], special_tokens=True)
    tokenizer.add_special_tokens({"pad_token": "<pad>"})
    llm.cfg.pad_token_id = llm.config.pad_token_id = tokenizer.pad_token_id
    llm.resize_token_embeddings(len(tokenizer))
    if args.peft:
        llm.enable_lora(r=16, alpha=32, dropout=0.05, seed=args.seed)
    return LLM(llm, args), tokenizer


def _inference(args, model, tokenizer, vocab, merges, data, device):
    """main.py:168-218: five seeded passes over the test split, per-seed result files, statistics over the seeds."""
    from .data_loader import DeviceBatchLoader, ECGTokenDataset
    from .file_utils import align_signal_text_files, sample_N_percent_from_lists
    from .model_utils import run_statistical_analysis
    from .runners import tester
    test_signals, test_texts = align_signal_text_files(f"{data}/ecg/test", f"{data}/text/test")
    if args.toy:
        test_signals, test_texts = sample_N_percent_from_lists(test_signals, test_texts, 0.25)
    test_data = ECGTokenDataset(test_signals, test_texts, vocab, merges, args=args, tokenizer=tokenizer)
    ckpt_dir = f"{args.runs_root}/{args.seed}/{args.checkpoint}"
    all_seed_results = []
    for seed in [0, 42, 123, 456, 789]:                                 # main.py:185-205
        random.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
        checkpoint = torch.load(f"{ckpt_dir}/best_model.pth", map_location=device)
        model.load_state_dict(checkpoint["model"])
        seed_results = tester(model, DeviceBatchLoader(test_data, batch_size=1, workers=args.workers), tokenizer, args)
        all_seed_results.append(seed_results)
        with open(f"{ckpt_dir}/seed_{seed}_results_{args.dataset}.json", "w") as f:
            json.dump({"averages": seed_results["metrics"], "qa_results": seed_results["qa_results"]}, f)
    stats_results = run_statistical_analysis(all_seed_results)
    with open(f"{ckpt_dir}/statistical_analysis_{args.dataset}.json", "w") as f:
        json.dump(stats_results, f)
    for metric, st in stats_results.items():
        print(f"\n{metric}:\nMean: {st['mean']:.2f}\nStd Dev: {st['std']:.2f}\n95% CI: [{st['conf_interval'][0]:.2f}, {st['conf_interval'][1]:.2f}]")
    print("Inference Complete")
    return stats_results


def init_distributed(args):
    """setup(), main.py:57-60 + 68-73: this process is rank RANK of WORLD_SIZE on GPU LOCAL_RANK; RCCL (backend "nccl") process group with
    the rendezvous of the environment.  Both launchers end here: torchrun sets the variables itself, `spawn_ranks` sets them per child."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("ECGB_MAIN_BACKEND", "nccl")                   # "gloo": the CPU tests of the launch path
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # this pool's driver supports dmabuf IPC only
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        args.device = torch.device(f"cuda:{local_rank}")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=args.device)      # RCCL
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _probe(rank, world, local_rank):
    """Test hook: one all-reduce over the group just built, then a JSON record of what this rank saw."""
    t = torch.tensor([float(rank + 1)])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    rec = {"rank": rank, "world": world, "local_rank": local_rank, "master_port": os.environ.get("MASTER_PORT"),
           "master_addr": os.environ.get("MASTER_ADDR"), "sum": float(t.item()), "pid": os.getpid(), "ppid": os.getppid()}
    with open(f"{os.environ['ECGB_MAIN_PROBE']}_{rank}.json", "w") as f:
        json.dump(rec, f)
    dist.destroy_process_group()
    return rec


def _spawned(rank, world, gpu_ids, port, argv):
    """Child of spawn_ranks: rank `rank` drives GPU gpu_ids[rank] (main.py:68-69) and meets the others on `port` (main.py:57-59)."""
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    os.environ["LOCAL_RANK"] = str(gpu_ids[rank])
    os.environ["MASTER_ADDR"] = "127.0.0.1"          # the reference says 'localhost'; the loopback address needs no resolver
    os.environ["MASTER_PORT"] = str(port)
    main(argv)


def spawn_ranks(args, argv=None):
    """mp.spawn(main, args=(world_size,), nprocs=world_size, join=True) of main.py:356-360: world size = number of ids in --gpus, start
    method "spawn", one child per listed GPU.  The parent has made no GPU call when it gets here (argument parsing only), so the
    children initialise HIP / RCCL in fresh processes; it waits for all of them and re-raises the first failure."""
    import sys
    import torch.multiprocessing as mp
    gpu_ids = [int(i) for i in args.gpus.split(",")]
    world = len(gpu_ids)
    child_argv = list(sys.argv[1:] if argv is None else argv)
    mp.spawn(_spawned, args=(world, gpu_ids, args.ports, child_argv), nprocs=world, join=True)
    return {"spawned": world, "gpus": gpu_ids, "port": args.ports}


def main(argv=None):
    from .data_loader import DeviceBatchLoader, ECGTokenDataset
    from .file_utils import align_signal_text_files, ensure_directory_exists, load_vocab_and_merges, sample_N_percent_from_lists
    from .model_utils import count_parameters, early_stopping, run_statistical_analysis
    from .parallel import GradAllReduce, shard_indices
    from .runners import tester, trainer, validater
    args = get_args(argv)
    rank, world = 0, 1
    if args.dis:
        if "RANK" not in os.environ:
            # the reference's own launch (main.py:356-360): `python main.py --dis --gpus 0,1,2,3 --ports P` spawns one process per listed GPU
            return spawn_ranks(args, argv)
        rank, world, local_rank = init_distributed(args)
        if os.environ.get("ECGB_MAIN_PROBE"):                               # test hook (tests/test_dist_cpu.py): rendezvous only
            return _probe(rank, world, local_rank)
    device = torch.device(args.device or "cuda:0")
    if device.type == "cuda":
        # every ctypes-launched kernel goes to torch's CURRENT stream of the CURRENT device: `--device cuda:N` must make N current
        # (the reference's .to(device) calls carry the device with the tensors, main.py:158; raw pointers do not)
        torch.cuda.set_device(device)
    if args.dev:
        args.epochs = 2
    gc.collect()
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    vocab, merges = load_vocab_and_merges(f"{args.data_root}/{args.tokenizer_check}.pkl")
    directory_path = run_directory(args)
    model, tokenizer = build_model_and_tokenizer(args, vocab, device)
    print(f"Total number of parameters: {count_parameters(model)}")
    data = f"{args.data_root}/{args.dataset}"

    if args.inference:
        try:
            return _inference(args, model, tokenizer, vocab, merges, data, device)
        finally:
            if args.dis:
                dist.destroy_process_group()

    train_signals, train_texts = align_signal_text_files(f"{data}/ecg/train", f"{data}/text/train")
    val_signals, val_texts = align_signal_text_files(f"{data}/ecg/val", f"{data}/text/val")
    if args.toy:
        train_signals, train_texts = sample_N_percent_from_lists(train_signals, train_texts, 0.25)
        val_signals, val_texts = sample_N_percent_from_lists(val_signals, val_texts, 0.25)
    training_data = ECGTokenDataset(train_signals, train_texts, vocab, merges, args=args, tokenizer=tokenizer)
    validation_data = ECGTokenDataset(val_signals, val_texts, vocab, merges, args=args, tokenizer=tokenizer)

    class _EpochSampler:                                                     # DistributedSampler(seed, shuffle=True), main.py:239-243
        def __init__(self, n):
            self.n, self.epoch = n, 0

        def set_epoch(self, e):
            self.epoch = e

        def __len__(self):
            return len(shard_indices(self.n, rank, world, shuffle=True, seed=args.seed, epoch=self.epoch))

        def __iter__(self):
            return iter(shard_indices(self.n, rank, world, shuffle=True, seed=args.seed, epoch=self.epoch))

    sampler = _EpochSampler(len(training_data)) if args.dis else None
    training_loader = DeviceBatchLoader(training_data, batch_size=args.batch_size, shuffle=not args.dis, sampler=sampler,
                                        seed=args.seed, workers=args.workers)
    validation_loader = DeviceBatchLoader(validation_data, batch_size=args.batch_size, workers=args.workers)
    optimizer = model.llm.make_optimizer(lr=args.lr, betas=(args.beta1, args.beta2), eps=args.eps, weight_decay=args.weight_decay,
                                         warmup=args.warmup)
    if args.dis:
        model.llm.grad_sync = GradAllReduce()
    if rank == 0:
        ensure_directory_exists(directory_path)
    train_loss, val_loss = [], []
    checkpoint = None
    try:
        for epoch in range(args.epochs):
            train_dic = trainer(model, training_loader, optimizer, args, epoch, directory_path)
            train_loss.append(train_dic["average_loss"])
            print(f"Training - Epoch: {epoch+1}\nTrain Loss: {train_dic['average_loss']}")
            val_dic = validater(model, validation_loader, args, epoch)
            val_loss.append(val_dic["average_loss"])
            print(f"Validating - Epoch: {epoch+1}\nVal Loss: {val_dic['average_loss']}")
            if early_stopping(val_loss, patience=args.patience, delta=0.01):
                print("Validation loss has stopped decreasing. Early stopping...")
                break
            checkpoint = {"model": model.state_dict(), "epoch": epoch}
            if val_dic["average_loss"] <= min(val_loss):                    # main.py:308-318
                if args.dis:
                    dist.barrier()
                if rank == 0:
                    torch.save(checkpoint, f"{directory_path}/best_model.pth")
                    print(f"Best model saved at epoch: {epoch+1}")
            print("-----------------------------------------------------------")
    except Exception as e:                                                   # main.py:321-333
        print(f"An error occurred: {e}")
        raise
    finally:
        # main.py:336-346: the latest-epoch checkpoint is ALWAYS written at exit (normal end, early stop or exception), so the
        # last epoch's weights survive when it was not the best one
        try:
            if checkpoint is not None and rank == 0:
                torch.save(checkpoint, f"{directory_path}/crash_model.pth")
                print("Final checkpoint saved as crash_model.pth")
        finally:
            if args.dis:
                dist.destroy_process_group()
    return {"train_loss": train_loss, "val_loss": val_loss, "directory": directory_path}


if __name__ == "__main__":
    main()
