"""Worker of test_gpu_decoder_model.py::test_data_parallel_gradients_two_ranks (not a test): rank r runs forward + backward
of the tiny golden decoder on its share of the golden batch with the data-parallel gradient exchange switched on, and
saves every parameter gradient.  Both ranks use cuda:0; the exchange runs over gloo (a one-GPU box cannot run RCCL
between two ranks)."""
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
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from ecg_byte_amd.parallel import GradAllReduce
    z, m = _load()
    m.grad_sync = GradAllReduce()
    batch = {k: v[rank_rows(rank)] for k, v in _batch(z).items()}
    m(**batch).loss.backward()
    torch.cuda.synchronize()
    np.savez(out_path + f".rank{rank}.npz", **{n: p.grad.float().cpu().numpy() for n, p in m.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
