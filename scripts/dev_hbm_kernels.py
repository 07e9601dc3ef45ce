"""Dev-only: the HBM-bound kernels of the train step alone at the C3 shapes (bench.py's hbm_kernel_report): ms per launch and fraction of 8 TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ecg_byte_amd.decoder import DecoderConfig
cfg = DecoderConfig.llama_3_2_1b(vocab_size=132515, pad_token_id=132514)
from ecg_byte_amd import decoder_ops as ops
from ecg_byte_amd import _lib
for cap, (rpw, bcap) in zip((4096, 16384, 65536, 1 << 20, 4096, 1 << 20), ((64, 2048), (16, 2048), (8, 4096), (4, 8192), (64, 2048), (4, 8192))):
    ops.set_stream_grid_cap(cap)
    _lib.lib().ecgb_set_rmsnorm_bwd_rows_per_wg(rpw); _lib.lib().ecgb_set_rmsnorm_bwd_grid_cap(bcap)
    print("stream grid cap", cap, " rmsnorm_bwd rows per workgroup", rpw, "cap", bcap)
    for r in bench.hbm_kernel_report(torch.device("cuda", 0), 32, 1024, cfg, 132515, reps=20):
        print(f"{r['kernel']:24s} {r['ms']*1e3:8.1f} us  {r['GB/s']:7.0f} GB/s  {r['frac']:.3f}")
    print()
