"""Worker of test_gpu_trainer.py::test_sharded_trainer_two_ranks_on_one_gpu (not a test): rank r trains on its slice of every case with
the HIP shard kernels (ecg_byte_amd.trainer.HipShard) and the real exchange loop; both ranks use cuda:0, the collectives run over
gloo (a one-GPU box cannot run RCCL between two ranks) -- `ddp_worker.py nccl` covers the RCCL side with one rank."""
import os
import pickle
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    cases_path, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from ecg_byte_amd.trainer import HipShard, bpe_train_sharded
    with open(cases_path, "rb") as f:
        cases = pickle.load(f)
    out = []
    for text, cuts, nm in cases:
        lo, hi = ([0] + cuts + [len(text)])[rank: rank + 2]
        t = torch.frombuffer(bytearray(text[lo:hi]), dtype=torch.uint8).cuda() if hi > lo else torch.empty(0, dtype=torch.uint8, device="cuda")
        out.append(bpe_train_sharded(HipShard(t, nm), nm))
    with open(out_path + f".rank{rank}", "wb") as f:
        pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
