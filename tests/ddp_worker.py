"""Worker of test_gpu_decoder_model.py::test_data_parallel_gradients_two_ranks and ::test_rccl_single_rank_gradient_exchange
(not a test): rank r runs forward + backward of the tiny golden decoder on its share of the golden batch with the
data-parallel gradient exchange switched on, and saves every parameter gradient and the number of collectives issued.

    ddp_worker.py OUT gloo    two ranks, both on cuda:0, exchange over gloo (a one-GPU box cannot run RCCL between two ranks)
    ddp_worker.py OUT nccl    ONE rank over RCCL (backend "nccl"): communicator init, asynchronous ncclAvg all-reduces of the
                              flat gradient buffer's buckets, stream ordering against the ctypes-launched kernels"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from test_gpu_decoder_model import _batch, _load   # noqa: E402


def rank_rows(rank):
    return [0, 1] if rank == 0 else [2]            # ragged on purpose: DDP averages the ranks' gradients, whatever their batch


def main():
    out_path = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend)
    from ecg_byte_amd.parallel import GradAllReduce
    z, m = _load()
    # 64 KB buckets: the tiny model's 0.6 MB of gradients leave in several collectives, as a real model's 25 MB buckets do
    m.grad_sync = GradAllReduce(bucket_bytes=64 << 10, single_rank_collectives=(world == 1))
    rows = rank_rows(rank) if world > 1 else [0, 1, 2]
    batch = {k: v[rows] for k, v in _batch(z).items()}
    for _ in range(2):                              # twice: the second backward reuses the flat buffer while nothing is pending
        for p in m.parameters():
            p.grad = None
        m(**batch).loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.float().cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}
    np.savez(out_path + f".rank{rank}.npz", __collectives__=np.int64(m.grad_sync.collectives), **grads)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
